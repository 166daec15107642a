// One frame of a JPEG XL codestream: LfGlobal, LF groups, HfGlobal, passes, pass groups -- everything the Java host
// parses before the transform stage (row f2). Counterpart of J/frame/{Frame,LFGlobal}.java, J/frame/group/*.java and
// J/frame/vardct/{HFBlockContext,LFChannelCorrelation,LFCoefficients (bitstream part),HFMetadata,HFGlobal (bitstream
// part),HFPass,HFCoefficients (bitstream part)}.java. The outputs are the boundary tensors of include/jxlatte_amd.h.
#pragma once
#include <array>
#include <memory>
#include <vector>

#include "headers.h"
#include "modular.h"

namespace jxf {

struct HFBlockContext {  // J/frame/vardct/HFBlockContext.java
    std::vector<uint8_t> cluster_map;
    int num_clusters = 15;
    std::vector<int32_t> lf_thresholds[3];
    std::vector<int32_t> qf_thresholds;
    int num_lf_contexts = 1;
    void read(BitReader& br);
};

struct PatchBlend { int mode = 0, alpha = 0; bool clamp = false; };
struct Patch {  // J/frame/features/Patch.java
    int ref = 0, x0 = 0, y0 = 0, w = 0, h = 0;
    std::vector<std::array<int32_t, 2>> positions;  // (y, x)
    std::vector<std::vector<PatchBlend>> blend;     // [position][1 + extra]
};

struct SplineData {  // J/frame/features/spline/SplinesBundle.java
    std::vector<int32_t> control;  // (y, x) pairs
    int32_t coeff[4][32];           // X, Y, B, sigma
};

struct QuantParams {  // one of the 17 parameter sets of J/frame/vardct/HFGlobal.java (DCTParams)
    int mode = 0;  // TransformType.MODE_*: 0 library default
    float denominator = 1.0f;
    std::vector<float> dct[3], par[3], p44[3];
};

struct LFGroupData {
    int cells_h = 0, cells_w = 0;  // LF group size in 8x8 cells
    int extra_precision = 0;
    bool has_lf_quant = false;
    Channel lf_quant[3];            // bitstream order Y, X, B (ModularStream channel order)
    std::vector<int32_t> lf_index;  // [cells_h * cells_w]
    // HF metadata
    int nb_blocks = 0;
    Channel x_from_y, b_from_y, sharpness;
    std::vector<uint8_t> dct_select;  // [cells_h * cells_w], 255 = unset
    std::vector<int32_t> hf_mul;
    std::vector<int32_t> block_yx;  // nb_blocks * 2
};

struct GroupCoeffs {
    int h[3] = {0, 0, 0}, w[3] = {0, 0, 0};
    std::vector<int32_t> q[3];
};

struct HFPass {
    uint32_t used_orders = 0;
    std::vector<uint16_t> order[13][3];  // packed (y << 8 | x)? no: pairs flattened y0,x0,y1,x1,...
    std::shared_ptr<EntropyCode> code;
};

// the part of a frame-level modular channel that sub-stream `idx` of group size `dim` carries, and the checked copy of the
// decoded part back into it (frame.cc; both validate untrusted geometry and throw BitstreamError)
Channel sub_channel(const Channel& full, int dim, int idx);
void copy_back(Channel& dst, const Channel& src, const Channel& want);

struct Frame {
    const ImageHeader* ih = nullptr;
    FrameHeader fh;
    Toc toc;
    int padded_w = 0, padded_h = 0;
    int group_cols = 0, lf_group_cols = 0, num_groups = 0, num_lf_groups = 0;
    // LfGlobal
    std::vector<Patch> patches;
    bool has_splines = false;
    int32_t spline_quant_adjust = 0;
    std::vector<SplineData> splines;
    bool has_noise = false;
    float noise[8] = {0};
    float lf_dequant[3] = {1.0f / 4096.0f, 1.0f / 512.0f, 1.0f / 256.0f};
    int global_scale = 0, quant_lf = 0;
    float scaled_dequant[3] = {0, 0, 0};
    HFBlockContext hfctx;
    int colour_factor = 84, x_factor_lf = 128, b_factor_lf = 128;
    float base_corr_x = 0.0f, base_corr_b = 1.0f;
    bool has_global_tree = false;
    MATree global_tree;
    ModularStream global_modular;
    // LF groups, HfGlobal, passes
    std::vector<LFGroupData> lf_groups;
    bool quant_all_default = true;
    QuantParams quant[17];
    int num_hf_presets = 1;
    std::vector<HFPass> hf_passes;
    std::vector<std::vector<GroupCoeffs>> coeffs;  // [pass][group]

    // parse the frame header + TOC at the reader position; section payloads follow
    void read_header(BitReader& br, const ImageHeader& image);
    // decode every section (br positioned right after the TOC); leaves br at the end of the frame data
    void decode(BitReader& br, const TransformHooks* hooks);
    size_t data_bytes() const;

    int colour_channels() const { return (ih->xyb_encoded || fh.encoding == kVarDCT) ? 3 : ih->colour_channels(); }

  private:
    BitReader section(const BitReader& base, size_t base_byte, int logical_index, bool single) const;
    void read_lf_global(BitReader& br);
    void read_lf_group(BitReader& br, int idx, std::vector<int>& replaced_idx);
    void read_hf_global(BitReader& br);
    void read_pass_group(BitReader& br, int pass, int group, const std::vector<int>& replaced_idx);
    void read_hf_coefficients(BitReader& br, int pass, int group);
    void read_quant_params(BitReader& br, int index);
    std::vector<size_t> offsets_;
    int pass_min_shift_[12] = {0}, pass_max_shift_[12] = {0};
};

}  // namespace jxf
