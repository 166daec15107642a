#include "modular.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace jxf {

void WPParams::read(BitReader& br) {  // WPParams.java:14-37
    if (br.flag()) return;
    p1 = (int)br.bits(5); p2 = (int)br.bits(5);
    p3a = (int)br.bits(5); p3b = (int)br.bits(5); p3c = (int)br.bits(5); p3d = (int)br.bits(5); p3e = (int)br.bits(5);
    for (int& v : w) v = (int)br.bits(4);
}

void Transform::read(BitReader& br) {  // TransformInfo.java:19-46
    tr = (int)br.bits(2);
    if (tr == 3) throw BitstreamError("Illegal Transform");
    begin_c = tr != kSqueeze ? (int)br.u32(0, 3, 8, 6, 72, 10, 1096, 13) : 0;
    rct_type = tr == kRCT ? (int)br.u32(6, 0, 0, 2, 2, 4, 10, 6) : 0;
    if (tr == kPalette) {
        num_c = (int)br.u32(1, 0, 3, 0, 4, 0, 1, 13);
        nb_colors = (int)br.u32(0, 8, 256, 10, 1280, 12, 5376, 16);
        nb_deltas = (int)br.u32(0, 0, 1, 8, 257, 10, 1281, 16);
        d_pred = (int)br.bits(4);
    }
    if (tr == kSqueeze) {
        const int n = (int)br.u32(0, 0, 1, 4, 9, 6, 41, 8);
        sp.resize(n);
        for (auto& s : sp) {  // SqueezeParam.java:10-15
            s.horizontal = br.flag();
            s.in_place = br.flag();
            s.begin_c = (int)br.u32(0, 3, 8, 6, 72, 10, 1096, 13);
            s.num_c = (int)br.u32(1, 0, 2, 0, 3, 0, 4, 4);
        }
    }
}

void MATree::read(BitReader& br) {  // MATree.java:31-87
    auto tc = std::make_shared<EntropyCode>();
    tc->read(br, 6);
    EntropyDecoder dec(tc);
    nodes.clear();
    uses_wp = false;
    int ctx_id = 0;
    int64_t remaining = 1;
    while (remaining-- > 0) {
        if (nodes.size() > (1u << 20)) throw BitstreamError("Tree too large");
        const int property = (int)dec.read(br, 1) - 1;
        MANode n;
        if (property >= 0) {
            n.property = property;
            n.value = unpack_signed(dec.read(br, 0));
            n.left = (int)(nodes.size() + remaining + 1);
            n.right = n.left + 1;
            remaining += 2;
            if (property == 15) uses_wp = true;
        } else {
            n.property = -1;
            n.ctx = ctx_id++;
            n.predictor = (int)dec.read(br, 2);
            if (n.predictor > 13) throw BitstreamError("Invalid predictor value");
            n.offset = unpack_signed(dec.read(br, 3));
            const uint32_t mul_log = dec.read(br, 4);
            if (mul_log > 30) throw BitstreamError("MulLog too large");
            const uint32_t mul_bits = dec.read(br, 5);
            if (mul_bits > (1u << (31 - mul_log)) - 2) throw BitstreamError("MulBits too large");
            n.multiplier = (mul_bits + 1) << mul_log;
            if (n.predictor == 6) uses_wp = true;
        }
        nodes.push_back(n);
    }
    dec.check_final("MA Tree");
    code = std::make_shared<EntropyCode>();
    code->read(br, (int)(nodes.size() + 1) / 2);
}

// ---- channel list surgery + stream header ------------------------------------------------------------------
void ModularStream::init(BitReader& br, std::vector<Channel> chans, int stream_index, const MATree* global_tree, int bit_depth) {
    stream_index_ = stream_index;
    bit_depth_ = bit_depth;
    channels = std::move(chans);
    transforms.clear();
    squeeze_steps.clear();
    nb_meta = 0;
    transformed_ = false;
    empty = channels.empty();
    if (empty) return;  // ModularStream.java:69-75: nothing is read for an empty channel list
    const bool use_global = br.flag();
    wp_ = WPParams();
    wp_.read(br);
    const int nb_tr = (int)br.u32(0, 0, 1, 0, 2, 4, 18, 8);
    transforms.resize(nb_tr);
    for (auto& t : transforms) t.read(br);
    squeeze_steps.resize(nb_tr);
    for (int i = 0; i < nb_tr; i++) {
        Transform& t = transforms[i];
        if (t.tr == Transform::kPalette) {  // :92-102
            if (t.begin_c < nb_meta) nb_meta += 2 - t.num_c;
            else nb_meta++;
            if (t.begin_c < 0 || t.begin_c + t.num_c > (int)channels.size() || t.num_c < 1) throw BitstreamError("Palette channel range");
            // the inverse recreates the removed channels as copies of the index channel: they must all have had its shape
            // (libjxl makes the same check; the Java reference runs into ArrayIndexOutOfBounds later)
            for (int j = t.begin_c + 1; j < t.begin_c + t.num_c; j++) {
                const Channel &a = channels[t.begin_c], &b = channels[j];
                if (a.w != b.w || a.h != b.h || a.hshift != b.hshift || a.vshift != b.vshift)
                    throw BitstreamError("Palette over channels of unequal size");
            }
            const int start = t.begin_c + 1;
            channels.erase(channels.begin() + start, channels.begin() + t.begin_c + t.num_c);
            if (t.nb_deltas > 0 && t.d_pred == 6) channels[t.begin_c].force_wp = true;
            channels.insert(channels.begin(), Channel(t.num_c, t.nb_colors, -1, -1));
        } else if (t.tr == Transform::kSqueeze) {  // :103-161
            std::vector<SqueezeStep> steps;
            if (t.sp.empty()) {
                const int first = nb_meta, count = (int)channels.size() - first;
                if (count < 1) throw BitstreamError("Squeeze without channels");
                int sw = channels[0].w, sh = channels[0].h;  // the reference sizes the default plan on channel 0
                if (count > 2 && channels[first + 1].w == sw && channels[first + 1].h == sh) {
                    steps.push_back({true, false, first + 1, 2});
                    steps.push_back({false, false, first + 1, 2});
                }
                if (sh >= sw && sh > 8) {
                    steps.push_back({false, true, first, count});
                    sh = (sh + 1) / 2;
                }
                while (sw > 8 || sh > 8) {
                    if (sw > 8) {
                        steps.push_back({true, true, first, count});
                        sw = (sw + 1) / 2;
                    }
                    if (sh > 8) {
                        steps.push_back({false, true, first, count});
                        sh = (sh + 1) / 2;
                    }
                }
            } else {
                steps = t.sp;
            }
            for (const SqueezeStep& s : steps) {
                const int begin = s.begin_c, end = begin + s.num_c - 1;
                if (begin < 0 || end >= (int)channels.size()) throw BitstreamError("Squeeze channel range");
                const int offset = s.in_place ? end + 1 : (int)channels.size();
                if (begin < nb_meta) {
                    if (!s.in_place) throw BitstreamError("squeeze meta must be in place");
                    if (end >= nb_meta) throw BitstreamError("squeeze meta must end in meta");
                    nb_meta += s.num_c;
                }
                for (int k = begin; k <= end; k++) {
                    Channel& ch = channels[k];
                    Channel res;
                    if (s.horizontal) {
                        const int w = ch.w;
                        ch.w = (w + 1) / 2;
                        ch.hshift++;
                        res = Channel(ch.h, w / 2, ch.vshift, ch.hshift);
                    } else {
                        const int h = ch.h;
                        ch.h = (h + 1) / 2;
                        ch.vshift++;
                        res = Channel(h / 2, ch.w, ch.vshift, ch.hshift);
                    }
                    res.ox = ch.ox; res.oy = ch.oy; res.force_wp = ch.force_wp;
                    channels.insert(channels.begin() + (offset + k - begin), res);
                }
            }
            squeeze_steps[i] = std::move(steps);
        } else if (t.tr == Transform::kRCT) {
            if (t.begin_c + 3 > (int)channels.size()) throw BitstreamError("RCT channel range");
        }
    }
    if (!use_global) {
        own_tree_.read(br);
        tree_ = &own_tree_;
    } else {
        if (!global_tree || global_tree->nodes.empty()) throw BitstreamError("Global MA tree requested but absent");
        tree_ = global_tree;
    }
    dec_.reset(tree_->code);
    dist_multiplier_ = 0;
    for (const Channel& c : channels) dist_multiplier_ = std::max<uint32_t>(dist_multiplier_, (uint32_t)c.w);
}

void ModularStream::decode_channels(BitReader& br, bool partial, int group_dim) {  // ModularStream.java:183-201
    if (empty) return;
    int channel_index = 0;
    for (size_t i = 0; i < channels.size(); i++) {
        Channel& ch = channels[i];
        if (partial && (int)i >= nb_meta && (ch.h > group_dim || ch.w > group_dim)) break;
        if (ch.w == 0 || ch.h == 0) {
            ch.allocate();
        } else {
            decode_channel(br, ch, channel_index);
            channel_index++;
        }
    }
    dec_.check_final("modular stream");
    if (!partial) apply_transforms(nullptr);
}

namespace {

inline int32_t iabs(int32_t v) { return v < 0 ? -v : v; }
inline int32_t clamp2(int32_t v, int32_t a, int32_t b) {
    const int32_t lo = a < b ? a : b, hi = a < b ? b : a;
    return v < lo ? lo : v > hi ? hi : v;
}
inline int32_t clamp3(int32_t v, int32_t a, int32_t b, int32_t c) {
    int32_t lo = a < b ? a : b, hi = a < b ? b : a;
    lo = lo < c ? lo : c;
    hi = hi > c ? hi : c;
    return v < lo ? lo : v > hi ? hi : v;
}

// neighbourhood accessors of ModularChannel.java:91-123 on a row-major plane
struct Nb {
    const int32_t* buf;
    int w;
    int32_t at(int x, int y) const { return buf[(size_t)y * w + x]; }
    int32_t W(int x, int y) const { return x > 0 ? at(x - 1, y) : y > 0 ? at(x, y - 1) : 0; }
    int32_t N(int x, int y) const { return y > 0 ? at(x, y - 1) : x > 0 ? at(x - 1, y) : 0; }
    int32_t NW(int x, int y) const { return x > 0 ? (y > 0 ? at(x - 1, y - 1) : at(x - 1, y)) : (y > 0 ? at(x, y - 1) : 0); }
    int32_t NE(int x, int y) const { return x + 1 < w && y > 0 ? at(x + 1, y - 1) : N(x, y); }
    int32_t NN(int x, int y) const { return y > 1 ? at(x, y - 2) : N(x, y); }
    int32_t NEE(int x, int y) const { return x + 2 < w && y > 0 ? at(x + 2, y - 1) : NE(x, y); }
    int32_t WW(int x, int y) const { return x > 1 ? at(x - 2, y) : W(x, y); }
};

// ModularChannel.prediction (:124-160); wp_pred = pred[y][x] of the weighted predictor
inline int32_t predict(const Nb& b, int x, int y, int k, int32_t wp_pred) {
    switch (k) {
        case 0: return 0;
        case 1: return b.W(x, y);
        case 2: return b.N(x, y);
        case 3: return (b.W(x, y) + b.N(x, y)) / 2;
        case 4: {
            const int32_t w = b.W(x, y), n = b.N(x, y), nw = b.NW(x, y);
            return iabs(n - nw) < iabs(w - nw) ? w : n;
        }
        case 5: {
            const int32_t w = b.W(x, y), n = b.N(x, y);
            return clamp2(w + n - b.NW(x, y), n, w);
        }
        case 6: return (wp_pred + 3) >> 3;
        case 7: return b.NE(x, y);
        case 8: return b.NW(x, y);
        case 9: return b.WW(x, y);
        case 10: return (b.W(x, y) + b.NW(x, y)) / 2;
        case 11: return (b.N(x, y) + b.NW(x, y)) / 2;
        case 12: return (b.N(x, y) + b.NE(x, y)) / 2;
        case 13:
            return (6 * b.N(x, y) - 2 * b.NN(x, y) + 7 * b.W(x, y) + b.WW(x, y) + b.NEE(x, y) + 3 * b.NE(x, y) + 8) / 16;
        default: throw BitstreamError("Invalid predictor value");
    }
}

struct DivLut {
    uint32_t v[64];
    DivLut() {
        for (int i = 0; i < 64; i++) v[i] = (1u << 24) / (uint32_t)(i + 1);
    }
};
const DivLut kOneL24OverKP1;

// self-correcting weighted predictor state (ModularChannel.prePredictWP, :161-216): errors of the four sub-predictors
// and of the final prediction, two rows each
struct WPState {
    int w = 0;
    std::vector<int32_t> err[5];  // [2 * w]: row y & 1
    int32_t subpred[4] = {0, 0, 0, 0};
    int32_t pred = 0;
    void init(int width) {
        w = width;
        for (auto& e : err) e.assign((size_t)2 * w, 0);
    }
    int32_t eW(int e, int x, int y) const { return x > 0 ? err[e][(size_t)(y & 1) * w + x - 1] : 0; }
    int32_t eN(int e, int x, int y) const { return y > 0 ? err[e][(size_t)((y - 1) & 1) * w + x] : 0; }
    int32_t eWW(int e, int x, int y) const { return x > 1 ? err[e][(size_t)(y & 1) * w + x - 2] : 0; }
    int32_t eNW(int e, int x, int y) const { return x > 0 && y > 0 ? err[e][(size_t)((y - 1) & 1) * w + x - 1] : eN(e, x, y); }
    int32_t eNE(int e, int x, int y) const { return x + 1 < w && y > 0 ? err[e][(size_t)((y - 1) & 1) * w + x + 1] : eN(e, x, y); }

    int32_t pre_predict(const WPParams& p, const Nb& b, int x, int y) {
        const int32_t n3 = b.N(x, y) << 3, nw3 = b.NW(x, y) << 3, ne3 = b.NE(x, y) << 3, w3 = b.W(x, y) << 3, nn3 = b.NN(x, y) << 3;
        const int32_t tN = eN(4, x, y), tW = eW(4, x, y), tNE = eNE(4, x, y), tNW = eNW(4, x, y);
        subpred[0] = w3 + ne3 - n3;
        subpred[1] = n3 - (((tW + tN + tNE) * p.p1) >> 5);
        subpred[2] = w3 - (((tW + tN + tNW) * p.p2) >> 5);
        subpred[3] = n3 - ((tNW * p.p3a + tN * p.p3b + tNE * p.p3c + (nn3 - n3) * p.p3d + (nw3 - w3) * p.p3e) >> 5);
        uint32_t weight[4];
        uint32_t wsum = 0;
        for (int e = 0; e < 4; e++) {
            int32_t es32 = eN(e, x, y) + eW(e, x, y) + eNW(e, x, y) + eWW(e, x, y) + eNE(e, x, y);
            int64_t es = es32;
            if (x + 1 == w) es += eW(e, x, y);
            const uint64_t esum = (uint64_t)es & 0xffffffffull;
            int shift = floor_log1p(esum) - 5;
            if (shift < 0) shift = 0;
            weight[e] = 4 + (((uint32_t)p.w[e] * kOneL24OverKP1.v[esum >> shift]) >> shift);
            wsum += weight[e];
        }
        const int log_weight = floor_log1p((uint64_t)wsum - 1) - 4;
        wsum = 0;
        for (int e = 0; e < 4; e++) {
            weight[e] >>= log_weight;
            wsum += weight[e];
        }
        int64_t s = (int64_t)(wsum >> 1) - 1;
        for (int e = 0; e < 4; e++) s += (int32_t)((uint32_t)subpred[e] * weight[e]);
        pred = (int32_t)((s * (int64_t)kOneL24OverKP1.v[wsum - 1]) >> 24);
        if (((tN ^ tW) | (tN ^ tNW)) <= 0) pred = clamp3(pred, w3, n3, ne3);
        int32_t max_error = tW;
        if (iabs(tN) > iabs(max_error)) max_error = tN;
        if (iabs(tNW) > iabs(max_error)) max_error = tNW;
        if (iabs(tNE) > iabs(max_error)) max_error = tNE;
        return max_error;
    }
    void update(int x, int y, int32_t value) {  // ModularChannel.java:310-314
        const size_t i = (size_t)(y & 1) * w + x;
        for (int e = 0; e < 4; e++) err[e][i] = (iabs(subpred[e] - (value << 3)) + 3) >> 3;
        err[4][i] = pred - (value << 3);
    }
};

}  // namespace

void ModularStream::decode_channel(BitReader& br, Channel& ch, int channel_index) {  // ModularChannel.decode (:279-318)
    if (ch.decoded) throw std::logic_error("Channel decoded twice");
    ch.decoded = true;
    ch.allocate();
    const MATree& tree = *tree_;
    const bool use_wp = ch.force_wp || tree.uses_wp;
    WPState wp;
    if (use_wp) wp.init(ch.w);
    if (ch.force_wp) ch.pred.assign((size_t)ch.w * ch.h, 0);
    // resolve the decisions that are constant for this channel (properties 0 and 1): MATree.compactify
    std::vector<MANode> nodes;
    {
        std::vector<int> remap(tree.nodes.size(), -1);
        // iterative copy with constant folding
        struct Item { int src; int* slot; };
        std::vector<int> root_slot(1, 0);
        std::vector<std::pair<int, std::pair<int, int>>> stack;  // (src, (parent index in nodes, which child: 0 left 1 right, -1 root))
        stack.push_back({0, {-1, -1}});
        while (!stack.empty()) {
            auto it = stack.back();
            stack.pop_back();
            int src = it.first;
            for (;;) {  // fold constant decisions
                const MANode& n = tree.nodes[src];
                if (n.property == 0) src = channel_index > n.value ? n.left : n.right;
                else if (n.property == 1) src = stream_index_ > n.value ? n.left : n.right;
                else break;
            }
            const int idx = (int)nodes.size();
            nodes.push_back(tree.nodes[src]);
            if (it.second.first >= 0) {
                if (it.second.second == 0) nodes[it.second.first].left = idx;
                else nodes[it.second.first].right = idx;
            }
            if (tree.nodes[src].property >= 0) {
                stack.push_back({tree.nodes[src].right, {idx, 1}});
                stack.push_back({tree.nodes[src].left, {idx, 0}});
            }
        }
    }
    const bool single_leaf = nodes[0].property < 0;
    // previous channels with the same geometry, newest first (properties >= 16; ModularChannel.propertyExpand :242-268)
    std::vector<const Channel*> refs;
    {
        // the reference walks list positions channelIndex - 1 .. 0 (channelIndex counts decoded, i.e. non-empty, channels)
        for (int j = std::min(channel_index, (int)channels.size()) - 1; j >= 0; j--) {
            const Channel& o = channels[j];
            if (o.w == ch.w && o.h == ch.h && o.vshift == ch.vshift && o.hshift == ch.hshift) refs.push_back(&o);
        }
    }
    const Nb nb{ch.buf.data(), ch.w};
    for (int y = 0; y < ch.h; y++) {
        int32_t* row = ch.row(y);
        for (int x = 0; x < ch.w; x++) {
            int32_t max_error = 0;
            if (use_wp) max_error = wp.pre_predict(wp_, nb, x, y);
            const MANode* leaf = &nodes[0];
            if (!single_leaf) {
                while (leaf->property >= 0) {
                    int32_t v;
                    const int k = leaf->property;
                    switch (k) {
                        case 0: v = channel_index; break;
                        case 1: v = stream_index_; break;
                        case 2: v = y; break;
                        case 3: v = x; break;
                        case 4: v = iabs(nb.N(x, y)); break;
                        case 5: v = iabs(nb.W(x, y)); break;
                        case 6: v = nb.N(x, y); break;
                        case 7: v = nb.W(x, y); break;
                        case 8: v = x > 0 ? nb.W(x, y) - (nb.W(x - 1, y) + nb.N(x - 1, y) - nb.NW(x - 1, y)) : nb.W(x, y); break;
                        case 9: v = nb.W(x, y) + nb.N(x, y) - nb.NW(x, y); break;
                        case 10: v = nb.W(x, y) - nb.NW(x, y); break;
                        case 11: v = nb.NW(x, y) - nb.N(x, y); break;
                        case 12: v = nb.N(x, y) - nb.NE(x, y); break;
                        case 13: v = nb.N(x, y) - nb.NN(x, y); break;
                        case 14: v = nb.W(x, y) - nb.WW(x, y); break;
                        case 15: v = max_error; break;
                        default: {
                            v = 0;
                            const int r = (k - 16) / 4, which = (k - 16) % 4;
                            if (k - 16 < 4 * channel_index && r < (int)refs.size()) {
                                const Channel& o = *refs[r];
                                const int32_t rC = o.buf[(size_t)y * o.w + x];
                                if (which == 0) v = iabs(rC);
                                else if (which == 1) v = rC;
                                else {
                                    const int32_t rW = x > 0 ? o.buf[(size_t)y * o.w + x - 1] : 0;
                                    const int32_t rN = y > 0 ? o.buf[(size_t)(y - 1) * o.w + x] : rW;
                                    const int32_t rNW = x > 0 && y > 0 ? o.buf[(size_t)(y - 1) * o.w + x - 1] : rW;
                                    const int32_t rG = rC - clamp2(rW + rN - rNW, rN, rW);
                                    v = which == 2 ? iabs(rG) : rG;
                                }
                            }
                        }
                    }
                    leaf = &nodes[v > leaf->value ? leaf->left : leaf->right];
                }
            }
            const uint32_t sym = dec_.read(br, leaf->ctx, dist_multiplier_);
            const int32_t diff = (int32_t)((uint32_t)unpack_signed(sym) * leaf->multiplier + (uint32_t)leaf->offset);
            const int32_t value = diff + predict(nb, x, y, leaf->predictor, wp.pred);
            row[x] = value;
            if (use_wp) {
                wp.update(x, y, value);
                if (ch.force_wp) ch.pred[(size_t)y * ch.w + x] = wp.pred;
            }
        }
    }
}

// ---- inverse transforms ---------------------------------------------------------------------------------
namespace {

// ModularChannel.tendency (:23-47)
inline int32_t tendency(int32_t a, int32_t b, int32_t c) {
    if (a >= b && b >= c) {
        int32_t x = (4 * a - 3 * c - b + 6) / 12;
        const int32_t d = 2 * (a - b), e = 2 * (b - c);
        if ((x - (x & 1)) > d) x = d + 1;
        if ((x + (x & 1)) > e) x = e;
        return x;
    }
    if (a <= b && b <= c) {
        int32_t x = (4 * a - 3 * c - b - 6) / 12;
        const int32_t d = 2 * (a - b), e = 2 * (b - c);
        if ((x + (x & 1)) < d) x = d - 1;
        if ((x - (x & 1)) < e) x = e;
        return x;
    }
    return 0;
}

Channel unsqueeze_h(const Channel& a, const Channel& r) {  // ModularChannel.inverseHorizontalSqueeze (:361-387)
    Channel o(a.h, a.w + r.w, a.vshift, a.hshift - 1);
    if ((a.w != r.w && a.w != r.w + 1) || r.h != a.h) throw BitstreamError("Corrupted squeeze transform");
    o.ox = a.ox; o.oy = a.oy; o.decoded = true;
    o.allocate();
    for (int y = 0; y < o.h; y++) {
        const int32_t* av = a.row(y);
        const int32_t* rv = r.row(y);
        int32_t* ov = o.row(y);
        for (int x = 0; x < r.w; x++) {
            const int32_t avg = av[x], next = x + 1 < a.w ? av[x + 1] : avg, left = x > 0 ? ov[2 * x - 1] : avg;
            const int32_t diff = rv[x] + tendency(left, avg, next);
            const int32_t first = avg + diff / 2;
            ov[2 * x] = first;
            ov[2 * x + 1] = first - diff;
        }
        if (a.w > r.w) ov[2 * r.w] = av[r.w];
    }
    return o;
}

Channel unsqueeze_v(const Channel& a, const Channel& r) {  // inverseVerticalSqueeze (:389-413)
    Channel o(a.h + r.h, a.w, a.vshift - 1, a.hshift);
    if ((a.h != r.h && a.h != r.h + 1) || r.w != a.w) throw BitstreamError("Corrupted squeeze transform");
    o.ox = a.ox; o.oy = a.oy; o.decoded = true;
    o.allocate();
    for (int y = 0; y < r.h; y++) {
        const int32_t* av = a.row(y);
        const int32_t* nv = y + 1 < a.h ? a.row(y + 1) : av;
        const int32_t* rv = r.row(y);
        const int32_t* up = y > 0 ? o.row(2 * y - 1) : av;
        int32_t* o0 = o.row(2 * y);
        int32_t* o1 = o.row(2 * y + 1);
        for (int x = 0; x < a.w; x++) {
            const int32_t avg = av[x];
            const int32_t diff = rv[x] + tendency(up[x], avg, nv[x]);
            const int32_t first = avg + diff / 2;
            o0[x] = first;
            o1[x] = first - diff;
        }
    }
    if (a.h > r.h) memcpy(o.row(2 * r.h), a.row(r.h), sizeof(int32_t) * (size_t)a.w);
    return o;
}

const int16_t kDeltaPalette[72][3] = {  // ModularStream.java:19-32
    {0, 0, 0}, {4, 4, 4}, {11, 0, 0}, {0, 0, -13}, {0, -12, 0}, {-10, -10, -10}, {-18, -18, -18}, {-27, -27, -27}, {-18, -18, 0},
    {0, 0, -32}, {-32, 0, 0}, {-37, -37, -37}, {0, -32, -32}, {24, 24, 45}, {50, 50, 50}, {-45, -24, -24}, {-24, -45, -45},
    {0, -24, -24}, {-34, -34, 0}, {-24, 0, -24}, {-45, -45, -24}, {64, 64, 64}, {-32, 0, -32}, {0, -32, 0}, {-32, 0, 32},
    {-24, -45, -24}, {45, 24, 45}, {24, -24, -45}, {-45, -24, 24}, {80, 80, 80}, {64, 0, 0}, {0, 0, -64}, {0, -64, -64},
    {-24, -24, 45}, {96, 96, 96}, {64, 64, 0}, {45, -24, -24}, {34, -34, 0}, {112, 112, 112}, {24, -45, -45}, {45, 45, -24},
    {0, -32, 32}, {24, -24, 45}, {0, 96, 96}, {45, -24, 24}, {24, -45, -24}, {-24, -45, 24}, {0, -64, 0}, {96, 0, 0},
    {128, 128, 128}, {64, 0, 64}, {144, 144, 144}, {96, 96, 0}, {-36, -36, 36}, {45, -24, -45}, {45, -45, -24}, {0, 0, -96},
    {0, 128, 128}, {0, 96, 0}, {45, 24, -45}, {-128, 0, 0}, {24, -45, 24}, {-45, 24, -45}, {64, 0, -64}, {64, -64, -64},
    {96, 0, 96}, {45, -45, 24}, {24, 45, -45}, {64, 64, -64}, {128, 128, 0}, {0, 0, -128}, {-24, 45, -45},
};

}  // namespace

void inverse_squeeze_cpu(std::vector<Channel>& channels, const std::vector<SqueezeStep>& steps) {  // ModularStream.java:231-259
    for (int j = (int)steps.size() - 1; j >= 0; j--) {
        const SqueezeStep& s = steps[j];
        const int begin = s.begin_c, end = begin + s.num_c - 1;
        const int offset = s.in_place ? end + 1 : (int)channels.size() + begin - end - 1;
        if (begin < 0 || offset + (end - begin) >= (int)channels.size()) throw BitstreamError("Squeeze channel range");
        for (int c = begin; c <= end; c++) {
            const int r = offset + c - begin;
            channels[c] = s.horizontal ? unsqueeze_h(channels[c], channels[r]) : unsqueeze_v(channels[c], channels[r]);
        }
        channels.erase(channels.begin() + offset, channels.begin() + offset + (end - begin + 1));
    }
}

void inverse_rct_cpu(Channel* v[3], int rct_type) {  // ModularStream.java:260-326 (arithmetic part)
    const int type = rct_type % 7;
    const size_t n = (size_t)v[0]->w * v[0]->h;
    int32_t* a = v[0]->buf.data();
    int32_t* b = v[1]->buf.data();
    int32_t* c = v[2]->buf.data();
    for (size_t i = 0; i < n; i++) {
        switch (type) {
            case 0: break;
            case 1: c[i] += a[i]; break;
            case 2: b[i] += a[i]; break;
            case 3: c[i] += a[i]; b[i] += a[i]; break;
            case 4: b[i] += (a[i] + c[i]) >> 1; break;
            case 5: {
                const int32_t ac = a[i] + c[i];
                b[i] += (a[i] + ac) >> 1;
                c[i] = ac;
                break;
            }
            default: {
                const int32_t bb = b[i], cc = c[i];
                const int32_t tmp = a[i] - (cc >> 1);
                const int32_t f = tmp - (bb >> 1);
                a[i] = f + bb;
                b[i] = cc + tmp;
                c[i] = f;
            }
        }
    }
}

void ModularStream::apply_transforms(const TransformHooks* hooks) {  // ModularStream.java:224-380
    if (transformed_ || empty) return;
    transformed_ = true;
    static const int kPerm[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
    for (int i = (int)transforms.size() - 1; i >= 0; i--) {
        const Transform& t = transforms[i];
        if (t.tr == Transform::kSqueeze) {
            if (!(hooks && hooks->squeeze && hooks->squeeze(hooks->user, channels, squeeze_steps[i]))) {
                if (hooks && hooks->required) throw std::logic_error("frame-level inverse Squeeze needs the device hook");
                inverse_squeeze_cpu(channels, squeeze_steps[i]);
            }
        } else if (t.tr == Transform::kRCT) {
            const int start = t.begin_c;
            if (start + 3 > (int)channels.size()) throw BitstreamError("RCT channel range");
            Channel* v[3] = {&channels[start], &channels[start + 1], &channels[start + 2]};
            if (v[1]->w != v[0]->w || v[1]->h != v[0]->h || v[2]->w != v[1]->w || v[2]->h != v[1]->h)
                throw BitstreamError("RCT must be performed on three equal size channels");
            for (Channel* c : v) c->allocate();
            if (hooks && hooks->rct && hooks->rct(hooks->user, v, t.rct_type)) continue;  // hook leaves output order
            if (hooks && hooks->required) throw std::logic_error("frame-level inverse RCT needs the device hook");
            inverse_rct_cpu(v, t.rct_type);
            const int perm = t.rct_type / 7;
            Channel tmp[3] = {std::move(channels[start]), std::move(channels[start + 1]), std::move(channels[start + 2])};
            for (int j = 0; j < 3; j++) channels[start + kPerm[perm][j]] = std::move(tmp[j]);
        } else {  // palette (:327-378)
            const int first = t.begin_c + 1, last = t.begin_c + t.num_c;
            if (first >= (int)channels.size()) throw BitstreamError("Palette channel range");
            const Channel pal = channels[0];
            for (int j = first + 1; j <= last; j++) channels.insert(channels.begin() + j, channels[first]);
            const int bd = bit_depth_;
            const Channel& fc = channels[first];
            const int H = fc.h, W = fc.w;
            for (int c = 0; c < t.num_c; c++) {
                Channel& ch = channels[first + c];
                ch.allocate();
                const Nb nb{ch.buf.data(), ch.w};
                for (int y = 0; y < H; y++)
                    for (int x = 0; x < W; x++) {
                        int32_t index = ch.buf[(size_t)y * W + x];
                        const bool is_delta = index < t.nb_deltas;
                        int32_t value;
                        if (index >= 0 && index < t.nb_colors) {
                            value = pal.buf[(size_t)c * pal.w + index];
                        } else if (index >= t.nb_colors) {
                            index -= t.nb_colors;
                            if (index < 64) {
                                value = ((index >> (2 * c)) % 4) * ((1 << bd) - 1) / 4 + (1 << std::max(0, bd - 3));
                            } else {
                                index -= 64;
                                for (int k = 0; k < c; k++) index /= 5;
                                value = (index % 5) * ((1 << bd) - 1) / 4;
                            }
                        } else if (c < 3) {
                            index = (-index - 1) % 143;
                            value = kDeltaPalette[(index + 1) >> 1][c];
                            if ((index & 1) == 0) value = -value;
                            if (bd > 8) value <<= std::min(bd, 24) - 8;
                        } else {
                            value = 0;
                        }
                        ch.buf[(size_t)y * W + x] = value;
                        if (is_delta) {
                            const int32_t wp_pred = ch.pred.empty() ? 0 : ch.pred[(size_t)y * W + x];
                            ch.buf[(size_t)y * W + x] += predict(nb, x, y, t.d_pred, wp_pred);
                        }
                    }
            }
            channels.erase(channels.begin());
            if (t.begin_c < nb_meta) nb_meta -= 2 - t.num_c;
            else nb_meta--;
        }
    }
}

}  // namespace jxf
