// Image-level and frame-level headers of a JPEG XL codestream (row f2). Counterpart of J/bundle/*.java,
// J/color/{ColorEncodingBundle,ToneMapping,OpsinInverseMatrix}.java (bitstream part), J/frame/FrameHeader.java,
// J/frame/features/RestorationFilter.java and the TOC part of J/frame/Frame.java.
#pragma once
#include <string>
#include <vector>

#include "bits.h"
#include "entropy.h"

namespace jxf {

struct BitDepth {
    bool float_samples = false;
    int bits = 8, exp_bits = 0;
    void read(BitReader& br);
};

struct ExtraChannel {
    int type = 0;  // 0 alpha, 1 depth, 2 spot, 3 selection mask, 4 black, 5 CFA, 6 thermal, 15/16 (non-)optional
    BitDepth depth;
    int dim_shift = 0;
    std::string name;
    bool alpha_associated = false;
    float spot[4] = {0, 0, 0, 0};
    int cfa_index = 1;
    void read(BitReader& br);
};

struct ColourEncoding {
    bool use_icc = false;
    int colour_space = 0;  // 0 RGB, 1 gray, 2 XYB, 3 unknown
    int white_point = 1;   // 1 D65, 2 custom, 10 E, 11 DCI
    float white_xy[2] = {0.3127f, 0.329f};
    int primaries = 1;  // 1 sRGB, 2 custom, 9 BT.2100, 11 P3
    float prim_xy[6] = {0.639998686f, 0.330010138f, 0.300003784f, 0.600003357f, 0.150002046f, 0.059997204f};
    int tf = (1 << 24) + 13;  // gamma * 1e7 if < 2^24, else 2^24 + enum (13 = sRGB)
    int rendering_intent = 1;
    void read(BitReader& br);
};

struct ToneMapping {
    float intensity_target = 255.0f, min_nits = 0.0f, linear_below = 0.0f;
    bool relative_to_max_display = false;
    void read(BitReader& br);
};

struct OpsinInverse {  // bitstream part of J/color/OpsinInverseMatrix.java
    float matrix[9] = {11.031566901960783f, -9.866943921568629f, -0.16462299647058826f, -3.254147380392157f, 4.418770392156863f,
                       -0.16462299647058826f, -3.6588512862745097f, 2.7129230470588235f, 1.9459282392156863f};
    float opsin_bias[3] = {-0.0037930732552754493f, -0.0037930732552754493f, -0.0037930732552754493f};
    float quant_bias[3] = {0.945349926692846f, 0.9299455010825141f, 0.9500648966626564f};
    float quant_bias_numerator = 0.145f;
    void read(BitReader& br);
};

struct ImageHeader {
    int width = 0, height = 0, level = 5;
    int orientation = 1;
    int intrinsic_w = 0, intrinsic_h = 0, preview_w = 0, preview_h = 0;
    bool have_animation = false, have_timecodes = false;
    int tps_num = 0, tps_den = 0, num_loops = 0;
    BitDepth depth;
    bool modular_16bit = true;
    std::vector<ExtraChannel> extra;
    bool xyb_encoded = true;
    ColourEncoding colour;
    ToneMapping tone;
    OpsinInverse opsin;
    bool custom_up[3] = {false, false, false};
    std::vector<float> up_weights[3];  // 15 / 55 / 210 coefficients when custom_up[i]
    size_t icc_encoded_size = 0;
    int colour_channels() const { return colour.colour_space == 1 ? 1 : 3; }
    void read(BitReader& br, int level);
};

struct BlendInfo {
    int mode = 0, alpha_channel = 0, source = 0;
    bool clamp = false;
    void read(BitReader& br, bool extra, bool full_frame);
};

struct Passes {
    int num_passes = 1, num_ds = 0;
    int shift[11] = {0}, down_sample[5] = {1, 1, 1, 1, 1}, last_pass[5] = {0};
    void read(BitReader& br);
};

struct Restoration {  // J/frame/features/RestorationFilter.java
    bool gab = true;
    float gab1[3] = {0.115169525f, 0.115169525f, 0.115169525f}, gab2[3] = {0.061248592f, 0.061248592f, 0.061248592f};
    int epf_iters = 2;
    float sharp_lut[8];
    float channel_scale[3] = {40.0f, 5.0f, 3.5f};
    float quant_mul = 0.46f, pass0_sigma = 0.9f, pass2_sigma = 6.5f, border_sad_mul = 2.0f / 3.0f, sigma_modular = 1.0f;
    Restoration();
    void read(BitReader& br, int encoding);
};

enum { kRegularFrame = 0, kLFFrame = 1, kReferenceOnly = 2, kSkipProgressive = 3 };
enum { kVarDCT = 0, kModular = 1 };
enum : uint64_t { kNoise = 1, kPatches = 2, kSplines = 16, kUseLFFrame = 32, kSkipAdaptiveLFSmoothing = 128 };

struct FrameHeader {
    int type = 0, encoding = 0;
    uint64_t flags = 0;
    bool do_ycbcr = false;
    int jpeg_up_y[3] = {0, 0, 0}, jpeg_up_x[3] = {0, 0, 0};  // stored as the reference does: max - own (shift to apply)
    int upsampling = 1;
    std::vector<int> ec_upsampling;
    int group_size_shift = 1, group_dim = 256, xqm = 3, bqm = 2;
    Passes passes;
    int lf_level = 0;
    bool have_crop = false;
    int x0 = 0, y0 = 0, width = 0, height = 0;  // bounds, already divided by upsampling / lf level and padded for subsampling
    BlendInfo blend;
    std::vector<BlendInfo> ec_blend;
    uint32_t duration = 0, timecode = 0;
    bool is_last = true;
    int save_as_reference = 0;
    bool save_before_ct = false;
    std::string name;
    Restoration rf;
    void read(BitReader& br, const ImageHeader& ih);
};

void skip_extensions(BitReader& br);

// Frame.readPermutation (Frame.java:190-215): Lehmer code -> permutation
std::vector<uint32_t> read_permutation(BitReader& br, EntropyDecoder& dec, uint32_t size, uint32_t skip);

struct Toc {
    std::vector<uint32_t> lengths;      // in bitstream order
    std::vector<uint32_t> permutation;  // empty = identity; section logical index -> bitstream slot
    void read(BitReader& br, uint32_t entries);
};

}  // namespace jxf
