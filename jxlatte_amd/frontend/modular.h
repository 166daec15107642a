// Modular sub-bitstreams: MA tree, channel decoding (14 predictors incl. the self-correcting weighted one), and the
// transform list (RCT / palette / squeeze). Counterpart of J/frame/modular/{MATree,ModularChannel,ModularStream,
// TransformInfo,WPParams,SqueezeParam}.java. Host-side front-end (row f2).
//
// Inverse squeeze / RCT of the FRAME-LEVEL stream are the hot path: ModularStream::apply_transforms takes hooks so the
// caller routes them to the device library (jxl_modular_apply / jxl_stage_rct); the built-in CPU versions serve the small
// side streams (LF coefficients, HF metadata, raw quant tables) exactly as the Java host does.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "bits.h"
#include "entropy.h"

namespace jxf {

struct WPParams {
    int p1 = 16, p2 = 10, p3a = 7, p3b = 7, p3c = 7, p3d = 0, p3e = 0;
    int w[4] = {13, 12, 12, 12};
    void read(BitReader& br);
};

struct SqueezeStep {
    bool horizontal = false, in_place = false;
    int begin_c = 0, num_c = 0;
};

struct Transform {
    enum { kRCT = 0, kPalette = 1, kSqueeze = 2 };
    int tr = 0, begin_c = 0, rct_type = 0, num_c = 0, nb_colors = 0, nb_deltas = 0, d_pred = 0;
    std::vector<SqueezeStep> sp;  // as read; the effective (default-expanded) list is kept by the stream
    void read(BitReader& br);
};

struct Channel {
    int w = 0, h = 0, hshift = 0, vshift = 0;
    int ox = 0, oy = 0;  // origin inside the frame-level channel (group / LF-group sub-channels)
    bool decoded = false, force_wp = false;
    std::vector<int32_t> buf;   // h * w once allocated
    std::vector<int32_t> pred;  // weighted-predictor prediction plane, kept only when force_wp (delta palette, d_pred 6)
    Channel() = default;
    Channel(int h_, int w_, int vs, int hs) : w(w_), h(h_), hshift(hs), vshift(vs) {}
    void allocate() {
        if (buf.size() != (size_t)w * h) buf.assign((size_t)w * h, 0);
    }
    int32_t* row(int y) { return buf.data() + (size_t)y * w; }
    const int32_t* row(int y) const { return buf.data() + (size_t)y * w; }
};

struct MANode {
    int property = -1;  // < 0: leaf
    int32_t value = 0;
    int left = 0, right = 0;  // child indices (property > value ? left : right)
    int ctx = 0, predictor = 0;
    int32_t offset = 0;
    uint32_t multiplier = 1;
};

struct MATree {
    std::vector<MANode> nodes;
    std::shared_ptr<EntropyCode> code;  // histograms of the symbols this tree's contexts index
    bool uses_wp = false;
    void read(BitReader& br);
};

// hooks for the frame-level stream: return true when handled; channels are the stream's list at that moment
struct TransformHooks {
    void* user = nullptr;
    bool required = false;  // frame-level stream: a Squeeze / RCT without a handler is an error, never a CPU fallback
    // one Squeeze transform (all of its steps, already in application order): avg / residual channels in, merged out
    bool (*squeeze)(void* user, std::vector<Channel>& channels, const std::vector<SqueezeStep>& steps) = nullptr;
    bool (*rct)(void* user, Channel* v[3], int rct_type) = nullptr;
};

class ModularStream {
  public:
    // channel list given by the caller (shapes only); reads the stream header and replays the transforms' effect on
    // the channel list (ModularStream.java:66-178)
    void init(BitReader& br, std::vector<Channel> chans, int stream_index, const MATree* global_tree, int bit_depth);
    // ModularStream.decodeChannels; partial = stop at the first non-meta channel larger than group_dim (LfGlobal)
    void decode_channels(BitReader& br, bool partial, int group_dim);
    // ModularStream.applyTransforms (idempotent)
    void apply_transforms(const TransformHooks* hooks = nullptr);

    std::vector<Channel> channels;
    std::vector<Transform> transforms;
    std::vector<std::vector<SqueezeStep>> squeeze_steps;  // per transform index (effective list)
    int nb_meta = 0;
    bool empty = true;

  private:
    void decode_channel(BitReader& br, Channel& ch, int channel_index);
    int stream_index_ = 0, bit_depth_ = 8;
    uint32_t dist_multiplier_ = 1;
    bool transformed_ = false;
    WPParams wp_;
    MATree own_tree_;
    const MATree* tree_ = nullptr;
    EntropyDecoder dec_;
};

// CPU inverse steps (side streams; the frame-level stream goes through the hooks)
void inverse_squeeze_cpu(std::vector<Channel>& channels, const std::vector<SqueezeStep>& steps);
void inverse_rct_cpu(Channel* v[3], int rct_type);

}  // namespace jxf
