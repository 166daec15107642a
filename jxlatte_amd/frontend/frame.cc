#include "frame.h"

#include <algorithm>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>

#include "../../include/jxl_transform_types.h"

namespace jxf {

namespace {
const int kCMap[3] = {1, 0, 2};  // Frame.cMap (Frame.java:42): buffer index X,Y,B -> bitstream index

struct TT {
    const jxl_tt_info* t;
    int dsh() const { return t->ph >> 3; }
    int dsw() const { return t->pw >> 3; }
    bool flip() const { return jxl_tt_flip(t); }
};
inline TT tt_of(int type) { return TT{&JXL_TT[type]}; }
// first non-vertical type with that order id (TransformType.getByOrderID)
const jxl_tt_info* tt_by_order(int order) {
    for (const auto& t : JXL_TT)
        if (t.order_id == order && !(t.ph > t.pw)) return &t;
    return nullptr;
}
const jxl_tt_info* tt_by_param(int param) {
    for (const auto& t : JXL_TT)
        if (t.param_index == param && !(t.ph > t.pw)) return &t;
    return nullptr;
}
}  // namespace

void HFBlockContext::read(BitReader& br) {  // HFBlockContext.java:20-57
    if (br.flag()) {
        static const uint8_t kDefault[39] = {0, 1, 2, 2, 3, 3, 4, 5, 6, 6, 6, 6, 6, 7, 8, 9, 9, 10, 11, 12,
                                             13, 14, 14, 14, 14, 14, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14};
        cluster_map.assign(kDefault, kDefault + 39);
        num_clusters = 15;
        qf_thresholds.clear();
        for (auto& t : lf_thresholds) t.clear();
        num_lf_contexts = 1;
        return;
    }
    int nb_lf[3], lf_ctx = 1;
    for (int i = 0; i < 3; i++) {
        nb_lf[i] = (int)br.bits(4);
        lf_ctx *= nb_lf[i] + 1;
        lf_thresholds[i].resize(nb_lf[i]);
        for (auto& t : lf_thresholds[i]) t = unpack_signed(br.u32(0, 4, 16, 8, 272, 16, 65808, 32));
    }
    num_lf_contexts = lf_ctx;
    const int nb_qf = (int)br.bits(4);
    qf_thresholds.resize(nb_qf);
    for (auto& t : qf_thresholds) t = 1 + (int32_t)br.u32(0, 2, 4, 3, 12, 5, 44, 8);
    int bsize = 39 * (nb_qf + 1);
    for (int i = 0; i < 3; i++) bsize *= nb_lf[i] + 1;
    if (bsize > 39 * 64) throw BitstreamError("HF block Size too large");
    cluster_map.assign(bsize, 0);
    num_clusters = read_cluster_map(br, cluster_map, 16);
}

// geometry of the part of `full` that sub-stream number `idx` of group size `dim` carries (Frame.java:276-290, 322-331).
// Shifts come from squeeze bookkeeping on untrusted input: a shift that empties the group size would divide by zero in
// ceil_div (Java: ArithmeticException) and a shift >= 32 is undefined behaviour, so both are reported.
Channel sub_channel(const Channel& full, int dim, int idx) {
    Channel c(full.h, full.w, full.vshift, full.hshift);
    if (c.vshift < 0 || c.hshift < 0 || c.vshift > 30 || c.hshift > 30) throw BitstreamError("modular channel shift out of range");
    const int gh = dim >> c.vshift, gw = dim >> c.hshift;
    // the reference divides by the group WIDTH only (Frame.java:284, 329: ArithmeticException when it is 0); a group HEIGHT of 0
    // gives an empty sub-channel there (size.height = min(h, 0)), as in libjxl, and the stream decodes
    if (gw <= 0) throw BitstreamError("modular channel shift larger than the group size");
    if (gh <= 0) { c.h = c.w = 0; return c; }
    const int stride = ceil_div(c.w, gw);
    if (stride <= 0) { c.h = c.w = 0; return c; }
    c.oy = (idx / stride) * gh;
    c.ox = (idx % stride) * gw;
    c.h = std::min(c.h - c.oy, gh);
    c.w = std::min(c.w - c.ox, gw);
    if (c.h < 0 || c.w < 0) c.h = c.w = 0;
    return c;
}

// copy a decoded sub-channel back into the frame-level channel. The sub-stream's own transforms (palette over channels of
// unequal size, squeeze) can leave a channel with other dimensions or origin than requested; the reference then fails with
// ArrayIndexOutOfBoundsException -- here it is an invalid bitstream, never a write outside dst.buf.
void copy_back(Channel& dst, const Channel& src, const Channel& want) {
    dst.allocate();
    if (src.h != want.h || src.w != want.w || src.oy != want.oy || src.ox != want.ox)
        throw BitstreamError("modular sub-stream returned a channel of the wrong size");
    if (src.h == 0 || src.w == 0) return;
    if (src.oy < 0 || src.ox < 0 || (int64_t)src.oy + src.h > dst.h || (int64_t)src.ox + src.w > dst.w)
        throw BitstreamError("modular sub-stream channel leaves the frame channel");
    if (src.buf.size() < (size_t)src.h * src.w) throw BitstreamError("modular sub-stream channel was not decoded");
    for (int y = 0; y < src.h; y++) memcpy(dst.row(y + src.oy) + src.ox, src.row(y), sizeof(int32_t) * (size_t)src.w);
}

void Frame::read_header(BitReader& br, const ImageHeader& image) {  // Frame.readFrameHeader + readTOC
    ih = &image;
    br.align_to_byte();
    fh.read(br, image);
    // untrusted geometry: the crop size is a 30-bit field per axis. Bound the frame exactly as the image is bounded (api.cc) so
    // that every product below fits an int and the TOC cannot ask for gigabytes (the Java reference dies with
    // OutOfMemoryError / NegativeArraySizeException here; this port reports it)
    if (fh.width <= 0 || fh.height <= 0) throw BitstreamError("empty frame");
    if ((int64_t)fh.width * fh.height > (1ll << 28)) throw UnsupportedError("frame larger than 2^28 pixels");
    group_cols = ceil_div(fh.width, fh.group_dim);
    lf_group_cols = ceil_div(fh.width, fh.group_dim << 3);
    num_groups = group_cols * ceil_div(fh.height, fh.group_dim);
    num_lf_groups = lf_group_cols * ceil_div(fh.height, fh.group_dim << 3);
    {  // Frame.getPaddedFrameSize (:915-933)
        const int fy = 1 << std::max({fh.jpeg_up_y[0], fh.jpeg_up_y[1], fh.jpeg_up_y[2]});
        const int fx = 1 << std::max({fh.jpeg_up_x[0], fh.jpeg_up_x[1], fh.jpeg_up_x[2]});
        int h = fh.encoding == kVarDCT ? (fh.height + 7) >> 3 : fh.height;
        int w = fh.encoding == kVarDCT ? (fh.width + 7) >> 3 : fh.width;
        h = ceil_div(h, fy);
        w = ceil_div(w, fx);
        padded_h = fh.encoding == kVarDCT ? (h * fy) << 3 : h * fy;
        padded_w = fh.encoding == kVarDCT ? (w * fx) << 3 : w * fx;
    }
    const int64_t entries64 = (num_groups == 1 && fh.passes.num_passes == 1)
                                  ? 1 : 1 + (int64_t)num_lf_groups + 1 + (int64_t)num_groups * fh.passes.num_passes;
    if (entries64 > (1ll << 22)) throw UnsupportedError("TOC with more than 2^22 entries");
    // every entry costs at least 10 bits of TOC: refuse counts the remaining bytes cannot hold before allocating for them
    if (entries64 * 10 > (int64_t)(br.size_bytes() - std::min(br.size_bytes(), br.byte_pos())) * 8 + 64)
        throw BitstreamError("TOC larger than the codestream");
    const uint32_t entries = (uint32_t)entries64;
    toc.read(br, entries);
    offsets_.assign(entries + 1, 0);
    for (uint32_t i = 0; i < entries; i++) offsets_[i + 1] = offsets_[i] + toc.lengths[i];
}

size_t Frame::data_bytes() const { return offsets_.empty() ? 0 : offsets_.back(); }

BitReader Frame::section(const BitReader& base, size_t base_byte, int logical_index, bool single) const {
    if (single) return base;
    const uint32_t slot = toc.permutation.empty() ? (uint32_t)logical_index : toc.permutation[logical_index];
    if (slot >= toc.lengths.size()) throw BitstreamError("TOC permutation out of range");
    const size_t begin = base_byte + offsets_[slot], len = toc.lengths[slot];
    if (begin + len > base.size_bytes()) throw BitstreamError("Unable to read full TOC entry");
    return BitReader(base.data() + begin, len);
}

// ---- LfGlobal (LFGlobal.java:30-103) -------------------------------------------------------------------------
void Frame::read_lf_global(BitReader& br) {
    const int extra = (int)ih->extra.size();
    if (fh.flags & kPatches) {
        auto code = std::make_shared<EntropyCode>();
        code->read(br, 10);
        EntropyDecoder dec(code);
        const uint32_t n = dec.read(br, 0);
        if (n > (1u << 24)) throw BitstreamError("That's a lot of patches!");
        int n_alpha = 0;
        for (const auto& e : ih->extra) n_alpha += e.type == 0;
        patches.resize(n);
        for (Patch& p : patches) {  // Patch.readPatch
            p.ref = (int)dec.read(br, 1);
            p.x0 = (int)dec.read(br, 3);
            p.y0 = (int)dec.read(br, 3);
            p.w = 1 + (int)dec.read(br, 2);
            p.h = 1 + (int)dec.read(br, 2);
            const int64_t count = 1 + (int64_t)dec.read(br, 7);
            if (count <= 0 || count > (1 << 24)) throw BitstreamError("That's a lot of patches!");
            p.positions.resize(count);
            p.blend.resize(count);
            for (int64_t j = 0; j < count; j++) {
                int32_t x, y;
                if (j == 0) {
                    x = (int32_t)dec.read(br, 4);
                    y = (int32_t)dec.read(br, 4);
                } else {
                    x = unpack_signed(dec.read(br, 6)) + p.positions[j - 1][1];
                    y = unpack_signed(dec.read(br, 6)) + p.positions[j - 1][0];
                }
                p.positions[j] = {y, x};
                p.blend[j].resize(extra + 1);
                for (PatchBlend& b : p.blend[j]) {
                    b.mode = (int)dec.read(br, 5);
                    if (b.mode >= 8) throw BitstreamError("Illegal blending mode in patch");
                    if (b.mode > 3 && n_alpha > 1) {
                        b.alpha = (int)dec.read(br, 8);
                        if (b.alpha >= extra) throw BitstreamError("Alpha out of bounds");
                    }
                    if (b.mode > 2) b.clamp = dec.read(br, 9) != 0;
                }
            }
        }
        dec.check_final("patches");
    }
    if (fh.flags & kSplines) {
        if (ih->colour_channels() < 3) throw BitstreamError("Cannot do splines in grayscale");
        has_splines = true;
        // SplinesBundle.java:22-73
        auto code = std::make_shared<EntropyCode>();
        code->read(br, 6);
        EntropyDecoder dec(code);
        const int64_t n = 1 + (int64_t)dec.read(br, 2);
        if (n > (1 << 16)) throw BitstreamError("too many splines");
        splines.assign((size_t)n, SplineData());
        std::vector<int32_t> sy(n), sx(n);
        for (int64_t i = 0; i < n; i++) {
            int32_t x = (int32_t)dec.read(br, 1), y = (int32_t)dec.read(br, 1);
            if (i != 0) {
                x = unpack_signed((uint32_t)x) + sx[i - 1];
                y = unpack_signed((uint32_t)y) + sy[i - 1];
            }
            sx[i] = x;
            sy[i] = y;
        }
        spline_quant_adjust = unpack_signed(dec.read(br, 0));
        for (int64_t i = 0; i < n; i++) {
            SplineData& sp = splines[i];
            const int64_t cc = 1 + (int64_t)dec.read(br, 3);
            if (cc > (1 << 20)) throw BitstreamError("too many spline control points");
            std::vector<int32_t> dx(cc - 1), dy(cc - 1);
            for (int64_t j = 0; j + 1 < cc; j++) {
                dx[j] = unpack_signed(dec.read(br, 4));
                dy[j] = unpack_signed(dec.read(br, 4));
            }
            int32_t cy = sy[i], cx = sx[i], ddy = 0, ddx = 0;
            sp.control = {cy, cx};
            for (int64_t j = 1; j < cc; j++) {
                ddy += dy[j - 1];
                ddx += dx[j - 1];
                cy += ddy;
                cx += ddx;
                sp.control.push_back(cy);
                sp.control.push_back(cx);
            }
            for (int k = 0; k < 4; k++)
                for (int j = 0; j < 32; j++) sp.coeff[k][j] = unpack_signed(dec.read(br, 5));
        }
        dec.check_final("splines");
    }
    if (fh.flags & kNoise) {
        if (ih->colour_channels() < 3) throw BitstreamError("Cannot do noise in grayscale");
        has_noise = true;
        for (float& v : noise) v = (float)br.bits(10) / 1024.0f;
    }
    if (!br.flag())
        for (float& v : lf_dequant) v = br.f16() * (1.0f / 128.0f);
    if (fh.encoding == kVarDCT) {
        global_scale = (int)br.u32(1, 11, 2049, 11, 4097, 12, 8193, 16);
        quant_lf = (int)br.u32(16, 0, 1, 5, 1, 8, 1, 16);
        for (int i = 0; i < 3; i++) scaled_dequant[i] = (float)(1 << 16) * lf_dequant[i] / (float)(global_scale * quant_lf);
        hfctx.read(br);
        if (!br.flag()) {  // LFChannelCorrelation.read
            colour_factor = (int)br.u32(84, 0, 256, 0, 2, 8, 258, 16);
            base_corr_x = br.f16();
            base_corr_b = br.f16();
            x_factor_lf = (int)br.bits(8);
            b_factor_lf = (int)br.bits(8);
        }
    }
    has_global_tree = br.flag();
    if (has_global_tree) global_tree.read(br);
    int ec_start = 0;
    if (fh.encoding == kModular) ec_start = (!fh.do_ycbcr && !ih->xyb_encoded && ih->colour.colour_space == 1) ? 1 : 3;
    std::vector<Channel> chans;
    for (int i = 0; i < extra + ec_start; i++) {
        const int ds = i < ec_start ? 0 : ih->extra[i - ec_start].dim_shift;
        chans.emplace_back(fh.height, fh.width, ds, ds);  // the reference keeps full-size planes for shifted extra channels
    }
    global_modular.init(br, std::move(chans), 0, has_global_tree ? &global_tree : nullptr, ih->depth.bits);
    global_modular.decode_channels(br, true, fh.group_dim);
}

// ---- LF groups (LFGroup.java, LFCoefficients.java:25-75, HFMetadata.java:16-52,93-119) ------------------------
void Frame::read_lf_group(BitReader& br, int idx, std::vector<int>& replaced_idx) {
    LFGroupData& g = lf_groups[idx];
    const int lfg_dim = fh.group_dim << 3;
    const int gy = idx / lf_group_cols, gx = idx % lf_group_cols;
    const int ph = std::min(lfg_dim, padded_h - gy * lfg_dim), pw = std::min(lfg_dim, padded_w - gx * lfg_dim);
    g.cells_h = ph >> 3;
    g.cells_w = pw >> 3;
    if (fh.encoding == kVarDCT) {
        const bool adaptive = (fh.flags & (kSkipAdaptiveLFSmoothing | kUseLFFrame)) == 0;
        bool subsampled = false;
        for (int i = 0; i < 3; i++) subsampled |= fh.jpeg_up_y[i] != 0 || fh.jpeg_up_x[i] != 0;
        if (adaptive && subsampled) throw BitstreamError("Adaptive Smoothing is incompatible with subsampling");
        if (fh.flags & kUseLFFrame) {
            // LFCoefficients.java:44-57: the LF planes are copied from the LF frame's buffer by the caller (the host layer
            // holds lfBuffer[]); nothing is read here and lfIndex keeps its zero initialisation (:24, the early return)
            g.has_lf_quant = false;
            g.extra_precision = 0;
            g.lf_index.assign((size_t)g.cells_h * g.cells_w, 0);
        } else {
        g.extra_precision = (int)br.bits(2);
        std::vector<Channel> info(3);
        for (int i = 0; i < 3; i++)
            info[kCMap[i]] = Channel(g.cells_h >> fh.jpeg_up_y[i], g.cells_w >> fh.jpeg_up_x[i], fh.jpeg_up_y[i], fh.jpeg_up_x[i]);
        ModularStream ms;
        ms.init(br, std::move(info), 1 + idx, has_global_tree ? &global_tree : nullptr, ih->depth.bits);
        ms.decode_channels(br, false, fh.group_dim);
        if (ms.channels.size() != 3) throw BitstreamError("LF coefficient stream must end with three channels");
        for (int c = 0; c < 3; c++) g.lf_quant[c] = std::move(ms.channels[c]);
        g.has_lf_quant = true;
        // LFCoefficients.getLFIndex (:168-186)
        g.lf_index.assign((size_t)g.cells_h * g.cells_w, 0);
        for (int y = 0; y < g.cells_h; y++)
            for (int x = 0; x < g.cells_w; x++) {
                int index[3] = {0, 0, 0};
                for (int i = 0; i < 3; i++) {
                    const Channel& q = g.lf_quant[kCMap[i]];
                    const int sy = y >> fh.jpeg_up_y[i], sx = x >> fh.jpeg_up_x[i];
                    const int32_t v = q.buf[(size_t)sy * q.w + sx];
                    for (int32_t t : hfctx.lf_thresholds[i]) index[i] += v > t;
                }
                int li = index[0];
                li = li * ((int)hfctx.lf_thresholds[2].size() + 1) + index[2];
                li = li * ((int)hfctx.lf_thresholds[1].size() + 1) + index[1];
                g.lf_index[(size_t)y * g.cells_w + x] = li;
            }
        }
    }
    // modular channels of the frame-level stream carried by this LF group (Frame.decodeLFGroups :262-300)
    {
        std::vector<Channel> sub;
        for (int ci : replaced_idx) sub.push_back(sub_channel(global_modular.channels[ci], lfg_dim, idx));
        const std::vector<Channel> want = sub;
        ModularStream ms;
        ms.init(br, std::move(sub), 1 + num_lf_groups + idx, has_global_tree ? &global_tree : nullptr, ih->depth.bits);
        ms.decode_channels(br, false, fh.group_dim);
        if (ms.channels.size() < replaced_idx.size() && !ms.empty) throw BitstreamError("modular sub-stream lost channels");
        for (size_t j = 0; j < replaced_idx.size() && j < ms.channels.size(); j++)
            copy_back(global_modular.channels[replaced_idx[j]], ms.channels[j], want[j]);
    }
    if (fh.encoding != kVarDCT) return;
    // HFMetadata
    const int n = ceil_log2((uint64_t)g.cells_h * g.cells_w);
    g.nb_blocks = 1 + (int)br.bits(n);
    const int ch = (g.cells_h + 7) / 8, cw = (g.cells_w + 7) / 8;
    std::vector<Channel> info = {Channel(ch, cw, 0, 0), Channel(ch, cw, 0, 0), Channel(2, g.nb_blocks, 0, 0),
                                 Channel(g.cells_h, g.cells_w, 0, 0)};
    ModularStream ms;
    ms.init(br, std::move(info), 1 + 2 * num_lf_groups + idx, has_global_tree ? &global_tree : nullptr, ih->depth.bits);
    ms.decode_channels(br, false, fh.group_dim);
    if (ms.channels.size() != 4) throw BitstreamError("HF metadata stream must end with four channels");
    g.x_from_y = std::move(ms.channels[0]);
    g.b_from_y = std::move(ms.channels[1]);
    const Channel& info_ch = ms.channels[2];
    g.sharpness = std::move(ms.channels[3]);
    if (info_ch.h != 2 || info_ch.w != g.nb_blocks) throw BitstreamError("block info channel shape");
    g.dct_select.assign((size_t)g.cells_h * g.cells_w, 255);
    g.hf_mul.assign((size_t)g.cells_h * g.cells_w, 0);
    g.block_yx.resize((size_t)g.nb_blocks * 2);
    int ly = 0, lx = 0;
    for (int i = 0; i < g.nb_blocks; i++) {
        const int32_t type = info_ch.buf[i];
        if (type < 0 || type > 26) throw BitstreamError("Invalid Transform Type");
        const TT tt = tt_of(type);
        const int mul = 1 + info_ch.buf[(size_t)info_ch.w + i];
        // HFMetadata.placeBlock (:93-119): first free position at or after the last block, raster order
        bool placed = false;
        for (int y = ly, x = lx; y < g.cells_h && !placed; y++, x = 0) {
            for (; x < g.cells_w; x++) {
                if (tt.dsw() + x > g.cells_w) break;  // too wide for the rest of this row
                bool occupied = false;
                for (int ix = 0; ix < tt.dsw(); ix++) {
                    const uint8_t o = g.dct_select[(size_t)y * g.cells_w + x + ix];
                    if (o != 255) {
                        x += (tt_of(o).dsw()) - 1;
                        occupied = true;
                        break;
                    }
                }
                if (occupied) continue;
                if (y + tt.dsh() > g.cells_h) throw BitstreamError("Varblock leaves the LF group");
                for (int iy = 0; iy < tt.dsh(); iy++)
                    for (int ix = 0; ix < tt.dsw(); ix++) {
                        g.dct_select[(size_t)(y + iy) * g.cells_w + x + ix] = (uint8_t)type;
                        g.hf_mul[(size_t)(y + iy) * g.cells_w + x + ix] = mul;
                    }
                g.block_yx[(size_t)i * 2] = y;
                g.block_yx[(size_t)i * 2 + 1] = x;
                ly = y;
                lx = x;
                placed = true;
                break;
            }
        }
        if (!placed) throw BitstreamError("Could not find place for block");
    }
}

// ---- HfGlobal (HFGlobal.java:190-302) + passes (Pass.java, HFPass.java) ---------------------------------------
void Frame::read_quant_params(BitReader& br, int index) {
    QuantParams& q = quant[index];
    q.mode = (int)br.bits(3);
    auto read_dct_params = [&](std::vector<float> out[3]) {  // HFGlobal.readDCTParams
        const int n = 1 + (int)br.bits(4);
        for (int c = 0; c < 3; c++) {
            out[c].resize(n);
            for (float& v : out[c]) v = br.f16();
            out[c][0] *= 64.0f;
        }
    };
    if (!(q.mode == 0 || q.mode == 6 || q.mode == 7) && !((index >= 0 && index <= 3) || index == 9 || index == 10))
        throw BitstreamError("Invalid index for mode");  // TransformType.validateIndex
    switch (q.mode) {
        case 0: break;  // library default
        case 1:         // Hornuss
            for (int c = 0; c < 3; c++) {
                q.par[c].resize(3);
                for (float& v : q.par[c]) v = 64.0f * br.f16();
            }
            break;
        case 2:  // DCT2
            for (int c = 0; c < 3; c++) {
                q.par[c].resize(6);
                for (float& v : q.par[c]) v = 64.0f * br.f16();
            }
            break;
        case 3:  // DCT4
            for (int c = 0; c < 3; c++) {
                q.par[c].resize(2);
                for (float& v : q.par[c]) v = 64.0f * br.f16();
            }
            read_dct_params(q.dct);
            break;
        case 6: read_dct_params(q.dct); break;
        case 7: {  // raw: a modular image of the weights (no byte alignment before it, as the reference notes)
            q.denominator = br.f16();
            const jxl_tt_info* t = tt_by_param(index);
            const int mh = jxl_tt_mh(t), mw = jxl_tt_mw(t);
            std::vector<Channel> info = {Channel(mh, mw, 0, 0), Channel(mh, mw, 0, 0), Channel(mh, mw, 0, 0)};
            ModularStream ms;
            ms.init(br, std::move(info), 1 + 3 * num_lf_groups + index, has_global_tree ? &global_tree : nullptr, ih->depth.bits);
            ms.decode_channels(br, false, fh.group_dim);
            if (ms.channels.size() != 3) throw BitstreamError("raw quant table must have three channels");
            for (int c = 0; c < 3; c++) {
                q.par[c].resize((size_t)mh * mw);
                for (size_t i = 0; i < q.par[c].size(); i++) q.par[c][i] = (float)ms.channels[c].buf[i];
            }
            break;
        }
        case 4:  // DCT4x8
            for (int c = 0; c < 3; c++) q.par[c] = {br.f16()};
            read_dct_params(q.dct);
            break;
        case 5:  // AFV
            for (int c = 0; c < 3; c++) {
                q.par[c].resize(9);
                for (int x = 0; x < 9; x++) {
                    q.par[c][x] = br.f16();
                    if (x < 6) q.par[c][x] *= 64.0f;
                }
            }
            read_dct_params(q.dct);
            read_dct_params(q.p44);
            break;
        default: throw BitstreamError("quant table mode");
    }
}

namespace {
// HFPass.getNaturalOrder (:16-66): LLF corner first in raster order, then zig-zag over a virtual square
const std::vector<uint16_t>& natural_order(int order_id) {
    static std::vector<uint16_t> cache[13];
    if (!cache[order_id].empty()) return cache[order_id];
    const jxl_tt_info* t = tt_by_order(order_id);
    const int ph = t->ph, pw = t->pw, dsh = ph >> 3, dsw = pw >> 3, maxd = std::max(dsh, dsw);
    std::vector<std::pair<int, int>> pts;
    pts.reserve((size_t)ph * pw);
    for (int y = 0; y < ph; y++)
        for (int x = 0; x < pw; x++) pts.push_back({y, x});
    std::stable_sort(pts.begin(), pts.end(), [&](const std::pair<int, int>& a, const std::pair<int, int>& b) {
        const bool al = a.first < dsh && a.second < dsw, bl = b.first < dsh && b.second < dsw;
        if (al != bl) return al;
        if (al) return a.first != b.first ? a.first < b.first : a.second < b.second;
        const int asy = a.first * maxd / dsh, asx = a.second * maxd / dsw, bsy = b.first * maxd / dsh, bsx = b.second * maxd / dsw;
        const int ak1 = asy + asx, bk1 = bsy + bsx;
        if (ak1 != bk1) return ak1 < bk1;
        int ak2 = asx - asy, bk2 = bsx - bsy;
        if (ak1 & 1) ak2 = -ak2;
        if (bk1 & 1) bk2 = -bk2;
        return ak2 < bk2;
    });
    std::vector<uint16_t>& out = cache[order_id];
    out.reserve(pts.size() * 2);
    for (auto& p : pts) {
        out.push_back((uint16_t)p.first);
        out.push_back((uint16_t)p.second);
    }
    return out;
}
}  // namespace

void Frame::read_hf_global(BitReader& br) {
    if (fh.encoding == kVarDCT) {
        quant_all_default = br.flag();
        if (!quant_all_default)
            for (int i = 0; i < 17; i++) read_quant_params(br, i);
        num_hf_presets = 1 + (int)br.bits(ceil_log1p((uint64_t)num_groups - 1));
    }
    // Frame.decodePasses / Pass.java: shift window of each pass + HFPass
    hf_passes.resize(fh.passes.num_passes);
    for (int p = 0; p < fh.passes.num_passes; p++) {
        pass_max_shift_[p] = p > 0 ? pass_min_shift_[p - 1] : 3;
        int n = -1;
        for (int i = 0; i <= fh.passes.num_ds; i++)
            if (fh.passes.last_pass[i] == p) {
                n = i;
                break;
            }
        pass_min_shift_[p] = n >= 0 ? ceil_log1p((uint64_t)fh.passes.down_sample[n] - 1) : pass_max_shift_[p];
        if (fh.encoding != kVarDCT) continue;
        HFPass& hp = hf_passes[p];
        hp.used_orders = br.u32(0x5F, 0, 0x13, 0, 0, 0, 0, 13);
        std::shared_ptr<EntropyCode> ocode;
        EntropyDecoder odec;
        if (hp.used_orders) {
            ocode = std::make_shared<EntropyCode>();
            ocode->read(br, 8);
            odec.reset(ocode);
        }
        for (int b = 0; b < 13; b++) {
            const std::vector<uint16_t>& nat = natural_order(b);
            const uint32_t len = (uint32_t)nat.size() / 2;
            for (int c = 0; c < 3; c++) {
                if (hp.used_orders & (1u << b)) {
                    const std::vector<uint32_t> perm = read_permutation(br, odec, len, len / 64);
                    hp.order[b][c].resize(nat.size());
                    for (uint32_t i = 0; i < len; i++) {
                        hp.order[b][c][2 * i] = nat[2 * perm[i]];
                        hp.order[b][c][2 * i + 1] = nat[2 * perm[i] + 1];
                    }
                } else {
                    hp.order[b][c] = nat;
                }
            }
        }
        if (hp.used_orders) odec.check_final("HFPass permutations");
        hp.code = std::make_shared<EntropyCode>();
        hp.code->read(br, 495 * num_hf_presets * hfctx.num_clusters);
    }
}

// ---- HF coefficients (HFCoefficients.java:49-138, :206-236) ---------------------------------------------------
namespace {
const int8_t kCoeffFreqCtx[64] = {-1, 0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 15, 16, 16, 17, 17,
                                  18, 18, 19, 19, 20, 20, 21, 21, 22, 22, 23, 23, 23, 23, 24, 24, 24, 24, 25, 25, 25, 25,
                                  26, 26, 26, 26, 27, 27, 27, 27, 28, 28, 28, 28, 29, 29, 29, 29, 30, 30, 30, 30};
const int16_t kCoeffNumNonzeroCtx[64] = {-1,  0,   31,  62,  62,  93,  93,  93,  93,  123, 123, 123, 123, 152, 152, 152,
                                         152, 152, 152, 152, 152, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180,
                                         180, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206,
                                         206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206};
}

void Frame::read_hf_coefficients(BitReader& br, int pass, int group) {
    const int hf_preset = (int)br.bits(ceil_log1p((uint64_t)num_hf_presets - 1));
    const int gy = group / group_cols, gx = group % group_cols;
    const int lfg_idx = (gy >> 3) * lf_group_cols + (gx >> 3);
    const LFGroupData& lfg = lf_groups[lfg_idx];
    const int offset = 495 * hfctx.num_clusters * hf_preset;
    const int shift = fh.passes.shift[pass];
    const HFPass& hp = hf_passes[pass];
    const int gh = std::min(fh.group_dim, padded_h - gy * fh.group_dim), gw = std::min(fh.group_dim, padded_w - gx * fh.group_dim);
    GroupCoeffs& out = coeffs[pass][group];
    for (int c = 0; c < 3; c++) {
        out.h[c] = gh >> fh.jpeg_up_y[c];
        out.w[c] = gw >> fh.jpeg_up_x[c];
        out.q[c].assign((size_t)out.h[c] * out.w[c], 0);
    }
    int32_t non_zeroes[3][32][32];
    memset(non_zeroes, 0, sizeof non_zeroes);
    EntropyDecoder dec(hp.code);
    const int pos_y = (gy & 7) << 5, pos_x = (gx & 7) << 5;  // Frame.groupPosInLFGroup << 5: group origin in LF-group cells
    const int nq = (int)hfctx.qf_thresholds.size();
    for (int i = 0; i < lfg.nb_blocks; i++) {
        const int by = lfg.block_yx[(size_t)i * 2], bx = lfg.block_yx[(size_t)i * 2 + 1];
        const int group_y = by - pos_y, group_x = bx - pos_x;
        if (group_y < 0 || group_x < 0 || group_y >= 32 || group_x >= 32) continue;
        const size_t cell = (size_t)by * lfg.cells_w + bx;
        const TT tt = tt_of(lfg.dct_select[cell]);
        const bool flip = tt.flip();
        const int hf_mult = lfg.hf_mul[cell], lf_index = lfg.lf_index[cell];
        const int num_blocks = tt.dsh() * tt.dsw();
        const int order_id = tt.t->order_id;
        for (int ci = 0; ci < 3; ci++) {
            const int c = kCMap[ci];  // Y, X, B
            const int sgy = group_y >> fh.jpeg_up_y[c], sgx = group_x >> fh.jpeg_up_x[c];
            if (group_y != sgy << fh.jpeg_up_y[c] || group_x != sgx << fh.jpeg_up_x[c]) continue;  // subsampled away
            const int py = sgy << 3, px = sgx << 3;
            // getPredictedNonZeroes
            int predicted;
            if (sgx == 0 && sgy == 0) predicted = 32;
            else if (sgx == 0) predicted = non_zeroes[c][sgy - 1][0];
            else if (sgy == 0) predicted = non_zeroes[c][0][sgx - 1];
            else predicted = (non_zeroes[c][sgy - 1][sgx] + non_zeroes[c][sgy][sgx - 1] + 1) >> 1;
            // getBlockContext
            int idx = (c < 2 ? 1 - c : c) * 13 + order_id;
            idx *= nq + 1;
            for (int32_t t : hfctx.qf_thresholds) idx += hf_mult > t;
            idx *= hfctx.num_lf_contexts;
            const int block_ctx = hfctx.cluster_map[idx + lf_index];
            // getNonZeroContext
            if (predicted > 64) predicted = 64;
            const int nz_ctx = offset + (predicted < 8 ? block_ctx + hfctx.num_clusters * predicted
                                                       : block_ctx + hfctx.num_clusters * (4 + predicted / 2));
            int64_t non_zero = dec.read(br, nz_ctx);
            const int per_block = (int)((non_zero + num_blocks - 1) / num_blocks);
            for (int iy = 0; iy < tt.dsh(); iy++)
                for (int ix = 0; ix < tt.dsw(); ix++)
                    if (sgy + iy < 32 && sgx + ix < 32) non_zeroes[c][sgy + iy][sgx + ix] = per_block;
            if (non_zero <= 0) continue;
            const std::vector<uint16_t>& order = hp.order[order_id][c];
            const int order_size = (int)order.size() / 2;
            if (non_zero > order_size - num_blocks) throw BitstreamError("Illegal nonzero count");
            const int hist_ctx = offset + 458 * block_ctx + 37 * hfctx.num_clusters;
            int32_t* plane = out.q[c].data();
            const int pw_ = out.w[c], ph_ = out.h[c];
            uint32_t prev_u = 0;
            for (int k = 0; k < order_size - num_blocks; k++) {
                const int prev = k == 0 ? (non_zero > order_size / 16 ? 0 : 1) : (prev_u != 0 ? 1 : 0);
                // getCoefficientContext(k + numBlocks, nonZero, numBlocks, prev)
                const int nzb = (int)((non_zero + num_blocks - 1) / num_blocks), kb = (k + num_blocks) / num_blocks;
                if (nzb > 63) throw BitstreamError("Illegal nonzero count");
                const int ctx = hist_ctx + (kCoeffNumNonzeroCtx[nzb] + kCoeffFreqCtx[kb]) * 2 + prev;
                const uint32_t u = dec.read(br, ctx);
                prev_u = u;
                const int oy = order[2 * (size_t)(k + num_blocks)], ox = order[2 * (size_t)(k + num_blocks) + 1];
                const int y = (flip ? ox : oy) + py, x = (flip ? oy : ox) + px;
                if (y >= ph_ || x >= pw_) throw BitstreamError("coefficient outside the group");
                plane[(size_t)y * pw_ + x] = (int32_t)((uint32_t)unpack_signed(u) << shift);
                if (u != 0 && --non_zero == 0) break;
            }
            if (non_zero != 0) throw BitstreamError("Illegal final nonzero count");
        }
    }
    dec.check_final("PassGroup HF coefficients");
}

void Frame::read_pass_group(BitReader& br, int pass, int group, const std::vector<int>& replaced_idx) {  // PassGroup.java:50-63
    if (fh.encoding == kVarDCT) read_hf_coefficients(br, pass, group);
    std::vector<Channel> sub;
    for (int ci : replaced_idx) sub.push_back(sub_channel(global_modular.channels[ci], fh.group_dim, group));  // Frame.decodePassGroups :320-331
    const std::vector<Channel> want = sub;
    ModularStream ms;
    ms.init(br, std::move(sub), 18 + 3 * num_lf_groups + num_groups * pass + group, has_global_tree ? &global_tree : nullptr,
            ih->depth.bits);
    ms.decode_channels(br, false, fh.group_dim);
    if (ms.channels.size() < replaced_idx.size() && !ms.empty) throw BitstreamError("modular sub-stream lost channels");
    for (size_t j = 0; j < replaced_idx.size() && j < ms.channels.size(); j++)
        copy_back(global_modular.channels[replaced_idx[j]], ms.channels[j], want[j]);
}

void Frame::decode(BitReader& br, const TransformHooks* hooks) {  // Frame.decodeFrame (:376-461), front part
    const bool single = toc.lengths.size() == 1;
    const size_t base = br.byte_pos();
    const bool timing = getenv("JXF_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  [jxf] %-12s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    if (!single && (br.bit_pos() & 7)) throw std::logic_error("TOC must end byte aligned");
    BitReader shared = br;  // single-section frames read everything through one cursor
    auto sec = [&](int logical) -> BitReader { return section(br, base, logical, false); };

    // LfGlobal
    if (single) read_lf_global(shared);
    else {
        BitReader r = sec(0);
        read_lf_global(r);
    }
    lap("LfGlobal");
    // LF groups
    std::vector<int> lf_replaced;
    for (size_t i = 0; i < global_modular.channels.size(); i++) {
        const Channel& c = global_modular.channels[i];
        if (!c.decoded && c.vshift >= 3 && c.hshift >= 3) lf_replaced.push_back((int)i);
    }
    lf_groups.assign(num_lf_groups, LFGroupData());
    if (single) {
        for (int g = 0; g < num_lf_groups; g++) read_lf_group(shared, g, lf_replaced);
    } else {
        for (int ci : lf_replaced) global_modular.channels[ci].allocate();
        std::exception_ptr err;
#pragma omp parallel for schedule(dynamic)
        for (int g = 0; g < num_lf_groups; g++) {
            try {
                BitReader r = sec(1 + g);
                read_lf_group(r, g, lf_replaced);
            } catch (...) {
#pragma omp critical(jxf_err)
                if (!err) err = std::current_exception();
            }
        }
        if (err) std::rethrow_exception(err);
    }
    for (int ci : lf_replaced) global_modular.channels[ci].decoded = true;
    lap("LF groups");
    // HfGlobal + pass headers
    BitReader hfg = single ? shared : sec(1 + num_lf_groups);
    read_hf_global(single ? shared : hfg);
    lap("HfGlobal");
    // pass groups
    coeffs.assign(fh.passes.num_passes, std::vector<GroupCoeffs>(fh.encoding == kVarDCT ? num_groups : 0));
    for (int p = 0; p < fh.passes.num_passes; p++) {
        std::vector<int> replaced;  // Pass.java:31-40
        for (size_t i = 0; i < global_modular.channels.size(); i++) {
            const Channel& c = global_modular.channels[i];
            if (c.decoded) continue;
            const int m = std::min(c.vshift, c.hshift);
            if (pass_min_shift_[p] <= m && m < pass_max_shift_[p]) replaced.push_back((int)i);
        }
        if (single) {
            for (int g = 0; g < num_groups; g++) read_pass_group(shared, p, g, replaced);
        } else {
            // the sections of one pass are independent bitstreams over disjoint regions: decode them on all host cores
            for (int ci : replaced) global_modular.channels[ci].allocate();
            std::exception_ptr err;
#pragma omp parallel for schedule(dynamic)
            for (int g = 0; g < num_groups; g++) {
                try {
                    BitReader r = sec(2 + num_lf_groups + p * num_groups + g);
                    read_pass_group(r, p, g, replaced);
                } catch (...) {
#pragma omp critical(jxf_err)
                    if (!err) err = std::current_exception();
                }
            }
            if (err) std::rethrow_exception(err);
        }
        for (int ci : replaced) global_modular.channels[ci].decoded = true;
    }
    lap("pass groups");
    for (Channel& c : global_modular.channels) c.allocate();
    global_modular.apply_transforms(hooks);
    lap("transforms");
    if (single) {
        br = shared;
    } else {
        br = BitReader(br.data(), br.size_bytes());
        // advance past the frame payload
        const size_t end = base + data_bytes();
        if (end > br.size_bytes()) throw BitstreamError("Unable to read full TOC entry");
        BitReader adv(br.data() + end, br.size_bytes() - end);
        br = adv;
    }
}

}  // namespace jxf
