#include "headers.h"

#include <algorithm>
#include <cmath>

namespace jxf {

void BitDepth::read(BitReader& br) {  // BitDepthHeader.java:19-28
    float_samples = br.flag();
    if (float_samples) {
        bits = (int)br.u32(32, 0, 16, 0, 24, 0, 1, 6);
        exp_bits = 1 + (int)br.bits(4);
    } else {
        bits = (int)br.u32(8, 0, 10, 0, 12, 0, 1, 6);
        exp_bits = 0;
    }
}

void skip_extensions(BitReader& br) {  // Extensions.readExtensions
    const uint64_t key = br.u64();
    uint64_t len[64];
    for (int i = 0; i < 64; i++) len[i] = (key >> i) & 1 ? br.u64() : 0;
    for (int i = 0; i < 64; i++)
        for (uint64_t j = 0; j < len[i]; j++) br.bits(8);
}

static std::string read_name(BitReader& br) {
    const uint32_t n = br.u32(0, 0, 0, 4, 16, 5, 48, 10);
    std::string s(n, '\0');
    for (uint32_t i = 0; i < n; i++) s[i] = (char)br.bits(8);
    return s;
}

void ExtraChannel::read(BitReader& br) {  // ExtraChannelInfo.java:24-59
    const bool d_alpha = br.flag();
    if (!d_alpha) {
        type = (int)br.enum_();
        if (!((type >= 0 && type <= 6) || type == 15 || type == 16)) throw BitstreamError("Illegal extra channel type");
        depth.read(br);
        dim_shift = (int)br.u32(0, 0, 3, 0, 4, 0, 1, 3);
        name = read_name(br);
        alpha_associated = type == 0 && br.flag();
    }
    if (type == 2)
        for (float& v : spot) v = br.f16();
    if (type == 5) cfa_index = (int)br.u32(1, 0, 0, 2, 3, 4, 19, 8);
}

static void read_custom_xy(BitReader& br, float* xy) {  // CIEXY.readCustom
    for (int i = 0; i < 2; i++) {
        const uint32_t u = br.u32(0, 19, 524288, 19, 1048576, 20, 2097152, 21);
        xy[i] = (float)unpack_signed(u) * 1e-6f;
    }
}

void ColourEncoding::read(BitReader& br) {  // ColorEncodingBundle.java:31-83 + ColorFlags tables
    const bool all_default = br.flag();
    use_icc = all_default ? false : br.flag();
    colour_space = all_default ? 0 : (int)br.enum_();
    if (colour_space < 0 || colour_space > 3) throw BitstreamError("Invalid ColorSpace enum");
    white_point = (!all_default && !use_icc && colour_space != 2) ? (int)br.enum_() : 1;
    switch (white_point) {
        case 1: white_xy[0] = 0.3127f; white_xy[1] = 0.329f; break;
        case 2: read_custom_xy(br, white_xy); break;
        case 10: white_xy[0] = 1.0f / 3.0f; white_xy[1] = 1.0f / 3.0f; break;
        case 11: white_xy[0] = 0.314f; white_xy[1] = 0.351f; break;
        default: throw BitstreamError("Invalid WhitePoint enum");
    }
    primaries = (!all_default && !use_icc && colour_space != 2 && colour_space != 1) ? (int)br.enum_() : 1;
    static const float kSRGB[6] = {0.639998686f, 0.330010138f, 0.300003784f, 0.600003357f, 0.150002046f, 0.059997204f};
    static const float kBT2100[6] = {0.708f, 0.292f, 0.170f, 0.797f, 0.131f, 0.046f};
    static const float kP3[6] = {0.680f, 0.320f, 0.265f, 0.690f, 0.150f, 0.060f};
    switch (primaries) {
        case 1: std::copy(kSRGB, kSRGB + 6, prim_xy); break;
        case 2:
            for (int i = 0; i < 3; i++) read_custom_xy(br, prim_xy + 2 * i);
            break;
        case 9: std::copy(kBT2100, kBT2100 + 6, prim_xy); break;
        case 11: std::copy(kP3, kP3 + 6, prim_xy); break;
        default: throw BitstreamError("Invalid Primaries enum");
    }
    if (!all_default && !use_icc) {
        if (br.flag()) {
            tf = (int)br.bits(24);
        } else {
            const int e = (int)br.enum_();
            if (!(e == 1 || e == 2 || e == 8 || e == 13 || e == 16 || e == 17 || e == 18)) throw BitstreamError("Illegal transfer function");
            tf = (1 << 24) + e;
        }
        rendering_intent = (int)br.enum_();
        if (rendering_intent > 3) throw BitstreamError("Invalid RenderingIntent enum");
    } else {
        tf = (1 << 24) + 13;
        rendering_intent = 1;
    }
}

void ToneMapping::read(BitReader& br) {  // ToneMapping.java:21-43
    if (br.flag()) return;
    intensity_target = br.f16();
    if (intensity_target <= 0.0f) throw BitstreamError("Intensity Target must be positive");
    min_nits = br.f16();
    if (min_nits < 0.0f || min_nits > intensity_target) throw BitstreamError("Min Nits out of range");
    relative_to_max_display = br.flag();
    linear_below = br.f16();
    if (linear_below < 0.0f || (relative_to_max_display && linear_below > 1.0f)) throw BitstreamError("Linear Below out of range");
}

void OpsinInverse::read(BitReader& br) {  // OpsinInverseMatrix.java:52-75
    if (br.flag()) return;
    for (float& v : matrix) v = br.f16();
    for (float& v : opsin_bias) v = br.f16();
    for (float& v : quant_bias) v = br.f16();
    quant_bias_numerator = br.f16();
}

static int width_from_ratio(int ratio, int h) {  // ImageHeader.getWidthFromRatio
    switch (ratio) {
        case 1: return h;
        case 2: return (int)((int64_t)h * 6 / 5);
        case 3: return (int)((int64_t)h * 4 / 3);
        case 4: return (int)((int64_t)h * 3 / 2);
        case 5: return (int)((int64_t)h * 16 / 9);
        case 6: return (int)((int64_t)h * 5 / 4);
        default: return h * 2;
    }
}

static void read_size(BitReader& br, int level, int& h, int& w) {  // ImageHeader.readSizeHeader
    const bool div8 = br.flag();
    h = div8 ? (int)(1 + br.bits(5)) << 3 : (int)br.u32(1, 9, 1, 13, 1, 18, 1, 30);
    const int ratio = (int)br.bits(3);
    if (ratio) w = width_from_ratio(ratio, h);
    else w = div8 ? (int)(1 + br.bits(5)) << 3 : (int)br.u32(1, 9, 1, 13, 1, 18, 1, 30);
    const int64_t max_dim = level <= 5 ? 1ll << 18 : 1ll << 28, max_area = level <= 5 ? 1ll << 30 : 1ll << 40;
    if (w > max_dim || h > max_dim || (int64_t)w * h > max_area) throw BitstreamError("Width or height too large");
}

static void read_preview_size(BitReader& br, int& h, int& w) {  // ImageHeader.readPreviewHeader
    const bool div8 = br.flag();
    h = div8 ? (int)br.u32(16, 0, 32, 0, 1, 5, 33, 9) : (int)br.u32(1, 6, 65, 8, 321, 10, 1345, 12);
    const int ratio = (int)br.bits(3);
    if (ratio) w = width_from_ratio(ratio, h);
    else w = div8 ? (int)br.u32(16, 0, 32, 0, 1, 5, 33, 9) : (int)br.u32(1, 6, 65, 8, 321, 10, 1345, 12);
    if (w > 4096 || h > 4096) throw BitstreamError("preview too large");
}

// context of the ICC byte stream (ImageHeader.getICCContext): needed only to consume the stream correctly
static int icc_context(const std::vector<uint8_t>& buf, size_t i) {
    if (i <= 128) return 0;
    const int b1 = buf[i - 1], b2 = buf[i - 2];
    auto alpha = [](int b) { return (b >= 'a' && b <= 'z') || (b >= 'A' && b <= 'Z'); };
    auto numish = [](int b) { return (b >= '0' && b <= '9') || b == '.' || b == ','; };
    int p1, p2;
    if (alpha(b1)) p1 = 0;
    else if (numish(b1)) p1 = 1;
    else if (b1 <= 1) p1 = 2 + b1;
    else if (b1 < 16) p1 = 4;
    else if (b1 > 240 && b1 < 255) p1 = 5;
    else if (b1 == 255) p1 = 6;
    else p1 = 7;
    if (alpha(b2)) p2 = 0;
    else if (numish(b2)) p2 = 1;
    else if (b2 < 16) p2 = 2;
    else if (b2 > 240) p2 = 3;
    else p2 = 4;
    return 1 + p1 + 8 * p2;
}

void ImageHeader::read(BitReader& br, int lvl) {  // ImageHeader.read (ImageHeader.java:219-319)
    if (br.bits(16) != 0x0AFF) throw BitstreamError("Not a JXL Codestream: 0xFF0A magic mismatch");
    level = lvl;
    read_size(br, level, height, width);
    const bool all_default = br.flag();
    const bool extra_fields = all_default ? false : br.flag();
    if (extra_fields) {
        orientation = 1 + (int)br.bits(3);
        if (br.flag()) read_size(br, level, intrinsic_h, intrinsic_w);
        if (br.flag()) read_preview_size(br, preview_h, preview_w);
        if (br.flag()) {  // AnimationHeader.read
            have_animation = true;
            tps_num = (int)br.u32(100, 0, 1000, 0, 1, 10, 1, 30);
            tps_den = (int)br.u32(1, 0, 1001, 0, 1, 8, 1, 10);
            num_loops = (int)br.u32(0, 0, 0, 3, 0, 16, 0, 32);
            have_timecodes = br.flag();
        }
    }
    if (!all_default) {
        depth.read(br);
        modular_16bit = br.flag();
        const uint32_t n_extra = br.u32(0, 0, 1, 0, 2, 4, 1, 12);
        extra.resize(n_extra);
        for (auto& e : extra) e.read(br);
        xyb_encoded = br.flag();
        colour.read(br);
    }
    if (extra_fields) tone.read(br);
    if (!all_default) skip_extensions(br);
    const bool default_matrix = br.flag();
    if (!default_matrix && xyb_encoded) opsin.read(br);
    const int cw_mask = default_matrix ? 0 : (int)br.bits(3);
    static const int kUpLen[3] = {15, 55, 210};
    for (int i = 0; i < 3; i++)
        if (cw_mask & (1 << i)) {
            custom_up[i] = true;
            up_weights[i].resize(kUpLen[i]);
            for (float& v : up_weights[i]) v = br.f16();
        }
    if (colour.use_icc) {
        const uint64_t enc = br.u64();
        if (enc > (1u << 28)) throw BitstreamError("ICC profile too large");
        icc_encoded_size = (size_t)enc;
        auto code = std::make_shared<EntropyCode>();
        code->read(br, 41);
        EntropyDecoder dec(code);
        std::vector<uint8_t> buf(icc_encoded_size);
        for (size_t i = 0; i < icc_encoded_size; i++) buf[i] = (uint8_t)dec.read(br, icc_context(buf, i));
        dec.check_final("ICC Stream");
    }
    br.align_to_byte();
}

void BlendInfo::read(BitReader& br, bool extra, bool full_frame) {  // BlendingInfo.java:27-45
    mode = (int)br.u32(0, 0, 1, 0, 2, 0, 3, 2);
    alpha_channel = (extra && (mode == 2 || mode == 3)) ? (int)br.u32(0, 0, 1, 0, 2, 0, 3, 3) : 0;
    clamp = (extra && (mode == 2 || mode == 4 || mode == 3)) ? br.flag() : false;
    source = (mode != 0 || !full_frame) ? (int)br.bits(2) : 0;
}

void Passes::read(BitReader& br) {  // PassesInfo.java:24-42
    num_passes = (int)br.u32(1, 0, 2, 0, 3, 0, 4, 3);
    num_ds = num_passes != 1 ? (int)br.u32(0, 0, 1, 0, 2, 0, 3, 1) : 0;
    if (num_ds >= num_passes) throw BitstreamError("num_ds < num_passes violated");
    for (int i = 0; i < num_passes - 1; i++) shift[i] = (int)br.bits(2);
    shift[num_passes - 1] = 0;
    for (int i = 0; i < num_ds; i++) down_sample[i] = 1 << br.bits(2);
    for (int i = 0; i < num_ds; i++) last_pass[i] = (int)br.u32(0, 0, 1, 0, 2, 0, 0, 3);
    down_sample[num_ds] = 1;
    last_pass[num_ds] = num_passes - 1;
}

Restoration::Restoration() {
    for (int i = 0; i < 8; i++) sharp_lut[i] = (float)i / 7.0f;
    sharp_lut[7] = 1.0f;
    for (float& v : sharp_lut) v *= quant_mul;
}

void Restoration::read(BitReader& br, int encoding) {  // RestorationFilter.java:46-80
    for (int i = 0; i < 8; i++) sharp_lut[i] = (float)i / 7.0f;
    sharp_lut[7] = 1.0f;
    const bool all_default = br.flag();
    gab = all_default ? true : br.flag();
    if (!all_default && gab && br.flag())
        for (int i = 0; i < 3; i++) {
            gab1[i] = br.f16();
            gab2[i] = br.f16();
        }
    epf_iters = all_default ? 2 : (int)br.bits(2);
    if (!all_default && epf_iters > 0 && encoding == kVarDCT && br.flag())
        for (float& v : sharp_lut) v = br.f16();
    if (!all_default && epf_iters > 0 && br.flag()) {
        for (float& v : channel_scale) v = br.f16();
        br.bits(32);
    }
    const bool sigma_custom = !all_default && epf_iters > 0 ? br.flag() : false;
    quant_mul = sigma_custom && encoding == kVarDCT ? br.f16() : 0.46f;
    pass0_sigma = sigma_custom ? br.f16() : 0.9f;
    pass2_sigma = sigma_custom ? br.f16() : 6.5f;
    border_sad_mul = sigma_custom ? br.f16() : 2.0f / 3.0f;
    sigma_modular = !all_default && epf_iters > 0 && encoding == kModular ? br.f16() : 1.0f;
    if (!all_default) skip_extensions(br);
    for (float& v : sharp_lut) v *= quant_mul;
}

void FrameHeader::read(BitReader& br, const ImageHeader& ih) {  // FrameHeader.java:78-196
    const bool all_default = br.flag();
    type = all_default ? kRegularFrame : (int)br.bits(2);
    encoding = all_default ? kVarDCT : (int)br.bits(1);
    flags = all_default ? 0 : br.u64();
    do_ycbcr = (!all_default && !ih.xyb_encoded) ? br.flag() : false;
    int mode_y[3] = {0, 0, 0}, mode_x[3] = {0, 0, 0};
    if (do_ycbcr && !(flags & kUseLFFrame))
        for (int i = 0; i < 3; i++) {
            const int mode = (int)br.bits(2);
            mode_y[i] = mode == 1 || mode == 3;
            mode_x[i] = mode == 1 || mode == 2;
        }
    ec_upsampling.assign(ih.extra.size(), 1);
    if (!all_default && !(flags & kUseLFFrame)) {
        upsampling = 1 << br.bits(2);
        for (int& u : ec_upsampling) u = 1 << br.bits(2);
    }
    group_size_shift = encoding == kModular ? (int)br.bits(2) : 1;
    group_dim = 128 << group_size_shift;
    if (ih.xyb_encoded && encoding == kVarDCT) {
        if (!all_default) {
            xqm = (int)br.bits(3);
            bqm = (int)br.bits(3);
        } else {
            xqm = 3;
            bqm = 2;
        }
    } else {
        xqm = bqm = 2;
    }
    if (!all_default && type != kReferenceOnly) passes.read(br);
    lf_level = type == kLFFrame ? 1 + (int)br.bits(2) : 0;
    have_crop = (!all_default && type != kLFFrame) ? br.flag() : false;
    if (have_crop && type != kReferenceOnly) {
        x0 = unpack_signed(br.u32(0, 8, 256, 11, 2304, 14, 18688, 30));
        y0 = unpack_signed(br.u32(0, 8, 256, 11, 2304, 14, 18688, 30));
    }
    if (have_crop) {
        width = (int)br.u32(0, 8, 256, 11, 2304, 14, 18688, 30);
        height = (int)br.u32(0, 8, 256, 11, 2304, 14, 18688, 30);
    } else {
        width = ih.width;
        height = ih.height;
    }
    const bool normal = !all_default && (type == kRegularFrame || type == kSkipProgressive);
    const bool full_frame = y0 <= 0 && x0 <= 0 && y0 + height >= ih.height && x0 + width >= ih.width;
    height = ceil_div(height, upsampling);
    width = ceil_div(width, upsampling);
    height = ceil_div(height, 1 << (3 * lf_level));
    width = ceil_div(width, 1 << (3 * lf_level));
    ec_blend.assign(ih.extra.size(), BlendInfo());
    if (normal) {
        blend.read(br, !ih.extra.empty(), full_frame);
        for (auto& b : ec_blend) b.read(br, true, full_frame);
    }
    duration = normal && ih.have_animation ? br.u32(0, 0, 1, 0, 0, 8, 0, 32) : 0;
    timecode = normal && ih.have_animation && ih.have_timecodes ? br.bits(32) : 0;
    is_last = normal ? br.flag() : type == kRegularFrame;
    save_as_reference = !all_default && type != kLFFrame && !is_last ? (int)br.bits(2) : 0;
    save_before_ct = !all_default && (type == kReferenceOnly || (full_frame && (type == kRegularFrame || type == kSkipProgressive) &&
                                                                 (duration == 0 || save_as_reference != 0) && !is_last && blend.mode == 0))
                         ? br.flag() : false;
    if (!all_default) name = read_name(br);
    if (!all_default) rf.read(br, encoding);
    if (!all_default) skip_extensions(br);
    const int max_y = std::max({mode_y[0], mode_y[1], mode_y[2]}), max_x = std::max({mode_x[0], mode_x[1], mode_x[2]});
    height = ceil_div(height, 1 << max_y) << max_y;
    width = ceil_div(width, 1 << max_x) << max_x;
    for (int i = 0; i < 3; i++) {
        jpeg_up_y[i] = max_y - mode_y[i];
        jpeg_up_x[i] = max_x - mode_x[i];
    }
}

std::vector<uint32_t> read_permutation(BitReader& br, EntropyDecoder& dec, uint32_t size, uint32_t skip) {
    auto ctx = [](uint32_t x) { return std::min(7, ceil_log1p(x)); };
    const uint32_t end = dec.read(br, ctx(size));
    if (end > size - skip) throw BitstreamError("Illegal end value in lehmer sequence");
    std::vector<uint32_t> lehmer(size, 0);
    for (uint32_t i = skip; i < end + skip; i++) {
        lehmer[i] = dec.read(br, ctx(i > skip ? lehmer[i - 1] : 0));
        if (lehmer[i] >= size - i) throw BitstreamError("Illegal lehmer value in lehmer sequence");
    }
    std::vector<uint32_t> temp(size), perm(size);
    for (uint32_t i = 0; i < size; i++) temp[i] = i;
    for (uint32_t i = 0; i < size; i++) {
        perm[i] = temp[lehmer[i]];
        temp.erase(temp.begin() + lehmer[i]);
    }
    return perm;
}

void Toc::read(BitReader& br, uint32_t entries) {  // Frame.readTOC (Frame.java:143-178)
    permutation.clear();
    if (br.flag()) {
        auto code = std::make_shared<EntropyCode>();
        code->read(br, 8);
        EntropyDecoder dec(code);
        permutation = read_permutation(br, dec, entries, 0);
        dec.check_final("TOC permutation");
    }
    br.align_to_byte();
    lengths.resize(entries);
    for (auto& l : lengths) l = br.u32(0, 10, 1024, 14, 17408, 22, 4211712, 30);
    br.align_to_byte();
}

}  // namespace jxf
