// C API of the front-end (include/jxlatte_frontend.h): container demux, image header, frame iteration, views.
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/jxlatte_frontend.h"
#include "frame.h"

using namespace jxf;

struct jxf_dec {
    std::vector<uint8_t> codestream;
    int level = 5;
    ImageHeader ih;
    BitReader br;
    std::unique_ptr<Frame> frame;
    bool done = false, skipped_preview = false;
    std::string error;
    // scratch for views
    std::vector<float> qv[3];
    std::vector<int32_t> patch_pos, patch_blend;
};

namespace {

uint64_t be(const uint8_t* p, int n) {
    uint64_t v = 0;
    for (int i = 0; i < n; i++) v = (v << 8) | p[i];
    return v;
}

// J/io/Demuxer.java: ISO-BMFF container -> concatenated jxlc / jxlp payloads; a bare codestream passes through
void demux(const uint8_t* d, size_t n, std::vector<uint8_t>& out, int& level) {
    static const uint8_t kSig[12] = {0, 0, 0, 0x0C, 'J', 'X', 'L', ' ', 0x0D, 0x0A, 0x87, 0x0A};
    if (n < 12 || memcmp(d, kSig, 12) != 0) {
        out.assign(d, d + n);
        return;
    }
    size_t pos = 12;
    while (pos + 8 <= n) {
        uint64_t size = be(d + pos, 4);
        const uint32_t tag = (uint32_t)be(d + pos + 4, 4);
        size_t header = 8;
        if (size == 1) {
            if (pos + 16 > n) throw BitstreamError("Truncated extended size");
            size = be(d + pos + 8, 8);
            header = 16;
        }
        size_t payload_end;
        if (size == 0) payload_end = n;  // box runs to the end of the file
        else {
            if (size < header || pos + size > n) throw BitstreamError("Illegal box size");
            payload_end = pos + (size_t)size;
        }
        const uint8_t* p = d + pos + header;
        const size_t len = payload_end - (pos + header);
        if (tag == 0x6A786C6C) {  // jxll
            if (len != 1 || (p[0] != 5 && p[0] != 10)) throw BitstreamError("Invalid level");
            level = p[0];
        } else if (tag == 0x6A786C63) {  // jxlc
            out.insert(out.end(), p, p + len);
        } else if (tag == 0x6A786C70) {  // jxlp: 4-byte sequence number first
            if (len < 4) throw BitstreamError("Truncated sequence number");
            out.insert(out.end(), p + 4, p + len);
        }
        pos = payload_end;
    }
}

template <typename F>
int32_t guarded(jxf_dec* d, F&& f) {
    try {
        return f();
    } catch (const BitstreamError& e) {
        d->error = e.what();
        return JXF_ERR_BITSTREAM;
    } catch (const UnsupportedError& e) {
        d->error = e.what();
        return JXF_ERR_UNSUPPORTED;
    } catch (const std::invalid_argument& e) {
        d->error = e.what();
        return JXF_ERR_ARGUMENT;
    } catch (const std::exception& e) {
        d->error = e.what();
        return JXF_ERR_STATE;
    }
}

struct HookCtx {
    const jxf_hooks* h;
    std::string* error;
};

bool squeeze_hook(void* user, std::vector<Channel>& channels, const std::vector<SqueezeStep>& steps) {
    HookCtx* hc = (HookCtx*)user;
    if (!hc->h || !hc->h->squeeze) return false;
    // shapes after the inverse: replay ModularStream.java:231-259 on shapes only
    std::vector<Channel> shapes;
    for (const Channel& c : channels) {
        Channel s(c.h, c.w, c.vshift, c.hshift);
        s.ox = c.ox; s.oy = c.oy;
        shapes.push_back(s);
    }
    for (int j = (int)steps.size() - 1; j >= 0; j--) {
        const SqueezeStep& s = steps[j];
        const int begin = s.begin_c, end = begin + s.num_c - 1;
        const int offset = s.in_place ? end + 1 : (int)shapes.size() + begin - end - 1;
        if (begin < 0 || offset + (end - begin) >= (int)shapes.size()) throw BitstreamError("Squeeze channel range");
        for (int c = begin; c <= end; c++) {
            const Channel& r = shapes[offset + c - begin];
            if (s.horizontal) { shapes[c].w += r.w; shapes[c].hshift--; }
            else { shapes[c].h += r.h; shapes[c].vshift--; }
        }
        shapes.erase(shapes.begin() + offset, shapes.begin() + offset + (end - begin + 1));
    }
    std::vector<jxf_chan> in(channels.size()), out(shapes.size());
    for (size_t i = 0; i < channels.size(); i++) {
        channels[i].allocate();
        in[i] = jxf_chan{channels[i].w, channels[i].h, channels[i].hshift, channels[i].vshift, channels[i].buf.data()};
    }
    for (size_t i = 0; i < shapes.size(); i++) {
        shapes[i].allocate();
        shapes[i].decoded = true;
        out[i] = jxf_chan{shapes[i].w, shapes[i].h, shapes[i].hshift, shapes[i].vshift, shapes[i].buf.data()};
    }
    std::vector<jxf_squeeze_step> st(steps.size());
    for (size_t i = 0; i < steps.size(); i++) st[i] = jxf_squeeze_step{steps[i].horizontal, steps[i].in_place, steps[i].begin_c, steps[i].num_c};
    const int32_t rc = hc->h->squeeze(hc->h->user, in.data(), (int32_t)in.size(), st.data(), (int32_t)st.size(), out.data(), (int32_t)out.size());
    if (rc != 0) throw std::runtime_error("squeeze hook failed with status " + std::to_string(rc));
    channels = std::move(shapes);
    return true;
}

bool rct_hook(void* user, Channel* v[3], int rct_type) {
    HookCtx* hc = (HookCtx*)user;
    if (!hc->h || !hc->h->rct) return false;
    const int32_t rc = hc->h->rct(hc->h->user, v[0]->buf.data(), v[1]->buf.data(), v[2]->buf.data(), (int64_t)v[0]->w * v[0]->h, rct_type);
    if (rc != 0) throw std::runtime_error("rct hook failed with status " + std::to_string(rc));
    return true;
}

}  // namespace

extern "C" {

jxf_dec* jxf_open(const uint8_t* data, size_t size, char* err, size_t err_len) {
    auto d = std::make_unique<jxf_dec>();
    try {
        if (!data || size < 2) throw BitstreamError("empty input");
        demux(data, size, d->codestream, d->level);
        d->br = BitReader(d->codestream.data(), d->codestream.size());
        d->ih.read(d->br, d->level);
        if (d->ih.extra.size() > JXF_MAX_EXTRA) throw UnsupportedError("more than 16 extra channels");
        // every plane of a frame is materialised: refuse images whose planes alone would need tens of GB
        if ((int64_t)d->ih.width * d->ih.height > (1ll << 28)) throw UnsupportedError("image larger than 2^28 pixels");
    } catch (const std::exception& e) {
        if (err && err_len) snprintf(err, err_len, "%s", e.what());
        return nullptr;
    }
    return d.release();
}

void jxf_close(jxf_dec* d) { delete d; }
const char* jxf_last_error(const jxf_dec* d) { return d ? d->error.c_str() : "null decoder"; }

int32_t jxf_get_image_info(const jxf_dec* d, jxf_image_info* o) {
    if (!d || !o) return JXF_ERR_ARGUMENT;
    memset(o, 0, sizeof *o);
    const ImageHeader& h = d->ih;
    o->width = h.width; o->height = h.height; o->level = h.level; o->orientation = h.orientation;
    o->bits_per_sample = h.depth.bits; o->exp_bits = h.depth.exp_bits; o->modular_16bit = h.modular_16bit;
    o->num_extra = (int32_t)h.extra.size(); o->xyb_encoded = h.xyb_encoded;
    o->colour_space = h.colour.colour_space; o->white_point = h.colour.white_point; o->primaries = h.colour.primaries;
    o->transfer = h.colour.tf; o->rendering_intent = h.colour.rendering_intent; o->use_icc = h.colour.use_icc;
    memcpy(o->white_xy, h.colour.white_xy, sizeof o->white_xy);
    memcpy(o->prim_xy, h.colour.prim_xy, sizeof o->prim_xy);
    o->intensity_target = h.tone.intensity_target; o->min_nits = h.tone.min_nits; o->linear_below = h.tone.linear_below;
    o->relative_to_max_display = h.tone.relative_to_max_display;
    memcpy(o->opsin_matrix, h.opsin.matrix, sizeof o->opsin_matrix);
    memcpy(o->opsin_bias, h.opsin.opsin_bias, sizeof o->opsin_bias);
    memcpy(o->quant_bias, h.opsin.quant_bias, sizeof o->quant_bias);
    o->quant_bias_numerator = h.opsin.quant_bias_numerator;
    o->have_animation = h.have_animation; o->have_preview = h.preview_w > 0;
    for (int i = 0; i < 3; i++) o->custom_up[i] = h.custom_up[i];
    for (size_t i = 0; i < h.extra.size(); i++) {
        o->ec_type[i] = h.extra[i].type; o->ec_bits[i] = h.extra[i].depth.bits; o->ec_exp_bits[i] = h.extra[i].depth.exp_bits;
        o->ec_dim_shift[i] = h.extra[i].dim_shift; o->ec_alpha_associated[i] = h.extra[i].alpha_associated;
    }
    return JXF_OK;
}

int32_t jxf_get_up_weights(const jxf_dec* d, int32_t k, float* out, int32_t cap) {
    if (!d || k < 0 || k > 2 || !out) return JXF_ERR_ARGUMENT;
    const auto& w = d->ih.up_weights[k];
    if ((int32_t)w.size() > cap) return JXF_ERR_ARGUMENT;
    memcpy(out, w.data(), sizeof(float) * w.size());
    return (int32_t)w.size();
}

int32_t jxf_next_frame(jxf_dec* d, const jxf_hooks* hooks) {
    if (!d) return JXF_ERR_ARGUMENT;
    return guarded(d, [&]() -> int32_t {
        if (d->done || d->br.at_end()) return JXF_END;
        if (d->ih.preview_w > 0 && !d->skipped_preview) {  // JXLCodestreamDecoder.java:583-590: the preview frame is skipped
            Frame pv;
            pv.read_header(d->br, d->ih);
            d->br.align_to_byte();
            const size_t end = d->br.byte_pos() + pv.data_bytes();
            if (end > d->br.size_bytes()) throw BitstreamError("Truncated preview frame");
            d->br = BitReader(d->br.data() + end, d->br.size_bytes() - end);
            d->skipped_preview = true;
        }
        d->frame = std::make_unique<Frame>();
        Frame& f = *d->frame;
        f.read_header(d->br, d->ih);
        if ((int64_t)f.padded_w * f.padded_h > (1ll << 28)) throw UnsupportedError("frame larger than 2^28 pixels");
        HookCtx hc{hooks, &d->error};
        TransformHooks th;
        th.user = &hc;
        th.required = true;
        th.squeeze = squeeze_hook;
        th.rct = rct_hook;
        f.decode(d->br, &th);
        if (f.fh.is_last) d->done = true;
        return JXF_OK;
    });
}

int32_t jxf_get_frame_info(const jxf_dec* d, jxf_frame_info* o) {
    if (!d || !o || !d->frame) return JXF_ERR_STATE;
    memset(o, 0, sizeof *o);
    const Frame& f = *d->frame;
    const FrameHeader& h = f.fh;
    o->type = h.type; o->encoding = h.encoding; o->do_ycbcr = h.do_ycbcr; o->upsampling = h.upsampling; o->group_dim = h.group_dim;
    o->xqm = h.xqm; o->bqm = h.bqm; o->lf_level = h.lf_level; o->flags = h.flags;
    for (int i = 0; i < 3; i++) { o->jpeg_up_y[i] = h.jpeg_up_y[i]; o->jpeg_up_x[i] = h.jpeg_up_x[i]; }
    for (size_t i = 0; i < h.ec_upsampling.size(); i++) o->ec_upsampling[i] = h.ec_upsampling[i];
    o->num_passes = h.passes.num_passes;
    for (int i = 0; i < 11; i++) o->pass_shift[i] = h.passes.shift[i];
    o->x0 = h.x0; o->y0 = h.y0; o->width = h.width; o->height = h.height; o->padded_width = f.padded_w; o->padded_height = f.padded_h;
    o->blend_mode = h.blend.mode; o->blend_alpha = h.blend.alpha_channel; o->blend_clamp = h.blend.clamp; o->blend_source = h.blend.source;
    for (size_t i = 0; i < h.ec_blend.size(); i++) {
        o->ec_blend_mode[i] = h.ec_blend[i].mode; o->ec_blend_alpha[i] = h.ec_blend[i].alpha_channel;
        o->ec_blend_clamp[i] = h.ec_blend[i].clamp; o->ec_blend_source[i] = h.ec_blend[i].source;
    }
    o->duration = h.duration; o->is_last = h.is_last; o->save_as_reference = h.save_as_reference; o->save_before_ct = h.save_before_ct;
    o->gab = h.rf.gab; o->epf_iters = h.rf.epf_iters;
    memcpy(o->gab1, h.rf.gab1, sizeof o->gab1); memcpy(o->gab2, h.rf.gab2, sizeof o->gab2);
    memcpy(o->epf_sharp_lut, h.rf.sharp_lut, sizeof o->epf_sharp_lut);
    memcpy(o->epf_channel_scale, h.rf.channel_scale, sizeof o->epf_channel_scale);
    o->epf_pass0_sigma = h.rf.pass0_sigma; o->epf_pass2_sigma = h.rf.pass2_sigma; o->epf_border_sad_mul = h.rf.border_sad_mul;
    o->epf_sigma_modular = h.rf.sigma_modular;
    o->num_groups = f.num_groups; o->num_lf_groups = f.num_lf_groups; o->group_cols = f.group_cols; o->lf_group_cols = f.lf_group_cols;
    o->num_patches = (int32_t)f.patches.size(); o->has_splines = f.has_splines; o->has_noise = f.has_noise;
    memcpy(o->noise, f.noise, sizeof o->noise);
    memcpy(o->lf_dequant, f.lf_dequant, sizeof o->lf_dequant); memcpy(o->scaled_dequant, f.scaled_dequant, sizeof o->scaled_dequant);
    o->global_scale = f.global_scale; o->quant_lf = f.quant_lf;
    o->colour_factor = f.colour_factor; o->x_factor_lf = f.x_factor_lf; o->b_factor_lf = f.b_factor_lf;
    o->base_corr_x = f.base_corr_x; o->base_corr_b = f.base_corr_b;
    o->quant_all_default = f.quant_all_default; o->num_hf_presets = f.num_hf_presets;
    o->num_modular_channels = (int32_t)f.global_modular.channels.size();
    return JXF_OK;
}

int32_t jxf_get_lfgroup(const jxf_dec* d, int32_t idx, jxf_lfgroup_view* o) {
    if (!d || !o || !d->frame) return JXF_ERR_STATE;
    const Frame& f = *d->frame;
    if (idx < 0 || idx >= (int32_t)f.lf_groups.size()) return JXF_ERR_ARGUMENT;
    const LFGroupData& g = f.lf_groups[idx];
    memset(o, 0, sizeof *o);
    o->cells_h = g.cells_h; o->cells_w = g.cells_w; o->extra_precision = g.extra_precision; o->has_lf_quant = g.has_lf_quant;
    static const int kCMap[3] = {1, 0, 2};
    if (g.has_lf_quant)
        for (int i = 0; i < 3; i++) {
            const Channel& c = g.lf_quant[kCMap[i]];
            o->lf_quant[i] = c.buf.data(); o->lf_h[i] = c.h; o->lf_w[i] = c.w;
        }
    o->n_blocks = g.nb_blocks;
    o->dct_select = g.dct_select.data(); o->hf_mul = g.hf_mul.data(); o->sharpness = g.sharpness.buf.data();
    o->x_from_y = g.x_from_y.buf.data(); o->b_from_y = g.b_from_y.buf.data(); o->block_yx = g.block_yx.data();
    return JXF_OK;
}

int32_t jxf_get_coeffs(const jxf_dec* d, int32_t pass, int32_t group, jxf_coeff_view* o) {
    if (!d || !o || !d->frame) return JXF_ERR_STATE;
    const Frame& f = *d->frame;
    if (pass < 0 || pass >= (int32_t)f.coeffs.size() || group < 0 || group >= (int32_t)f.coeffs[pass].size()) return JXF_ERR_ARGUMENT;
    const GroupCoeffs& g = f.coeffs[pass][group];
    for (int c = 0; c < 3; c++) { o->q[c] = g.q[c].data(); o->h[c] = g.h[c]; o->w[c] = g.w[c]; }
    return JXF_OK;
}

int32_t jxf_get_quant_params(const jxf_dec* dc, int32_t index, jxf_quant_view* o) {
    jxf_dec* d = const_cast<jxf_dec*>(dc);
    if (!d || !o || !d->frame || index < 0 || index > 16) return JXF_ERR_ARGUMENT;
    const QuantParams& q = d->frame->quant[index];
    auto flat = [](const std::vector<float> v[3], std::vector<float>& out) {
        out.clear();
        for (int c = 0; c < 3; c++) out.insert(out.end(), v[c].begin(), v[c].end());
        return (int32_t)v[0].size();
    };
    o->mode = q.mode; o->denominator = q.denominator;
    o->n_dct = flat(q.dct, d->qv[0]); o->n_par = flat(q.par, d->qv[1]); o->n_p44 = flat(q.p44, d->qv[2]);
    o->dct = d->qv[0].data(); o->par = d->qv[1].data(); o->p44 = d->qv[2].data();
    return JXF_OK;
}

int32_t jxf_get_patch(const jxf_dec* dc, int32_t index, jxf_patch_view* o) {
    jxf_dec* d = const_cast<jxf_dec*>(dc);
    if (!d || !o || !d->frame || index < 0 || index >= (int32_t)d->frame->patches.size()) return JXF_ERR_ARGUMENT;
    const Patch& p = d->frame->patches[index];
    o->ref = p.ref; o->x0 = p.x0; o->y0 = p.y0; o->w = p.w; o->h = p.h;
    o->n_positions = (int32_t)p.positions.size();
    o->n_blend = p.blend.empty() ? 0 : (int32_t)p.blend[0].size();
    d->patch_pos.clear(); d->patch_blend.clear();
    for (const auto& q : p.positions) { d->patch_pos.push_back(q[0]); d->patch_pos.push_back(q[1]); }
    for (const auto& bl : p.blend)
        for (const PatchBlend& b : bl) { d->patch_blend.push_back(b.mode); d->patch_blend.push_back(b.alpha); d->patch_blend.push_back(b.clamp); }
    o->positions = d->patch_pos.data(); o->blend = d->patch_blend.data();
    return JXF_OK;
}

int32_t jxf_num_splines(const jxf_dec* d) { return d && d->frame ? (int32_t)d->frame->splines.size() : 0; }

int32_t jxf_get_spline(const jxf_dec* d, int32_t index, jxf_spline_view* o) {
    if (!d || !o || !d->frame || index < 0 || index >= (int32_t)d->frame->splines.size()) return JXF_ERR_ARGUMENT;
    const SplineData& sp = d->frame->splines[index];
    o->quant_adjust = d->frame->spline_quant_adjust;
    o->n_control = (int32_t)sp.control.size() / 2;
    o->control = sp.control.data();
    o->coeff = &sp.coeff[0][0];
    return JXF_OK;
}

int32_t jxf_get_modular_channel(const jxf_dec* d, int32_t index, jxf_chan* o) {
    if (!d || !o || !d->frame) return JXF_ERR_STATE;
    auto& ch = d->frame->global_modular.channels;
    if (index < 0 || index >= (int32_t)ch.size()) return JXF_ERR_ARGUMENT;
    Channel& c = const_cast<Channel&>(ch[index]);
    c.allocate();
    *o = jxf_chan{c.w, c.h, c.hshift, c.vshift, c.buf.data()};
    return JXF_OK;
}

}  // extern "C"
