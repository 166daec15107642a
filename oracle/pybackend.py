"""Oracle-backed backend for jxlatte_amd.decoder.JXLDecoder -- TEST INFRASTRUCTURE ONLY (see oracle/jxl_oracle.h).

Lets the CPU-only test-suite drive the decoder's host logic (frame assembly, blending bookkeeping, colour management,
PNG output) and the C++ front-end on real .jxl files without a GPU. The product path (DeviceBackend) never imports this.
"""
import numpy as np

from jxlatte_amd import abi
from oracle import pyoracle as orc

TF_PQ, TF_SRGB = (1 << 24) + 16, (1 << 24) + 13


class OracleBackend:
    def __init__(self, threads=8):
        self.threads = threads

    def close(self):
        pass

    def vardct(self, params, weights, woffs, lfgroups, groups):
        H, W = params.height, params.width
        sy, sx = list(params.jpeg_upsampling_y), list(params.jpeg_upsampling_x)
        sub = any(sy) or any(sx)
        # channel c lives in the first (H >> sy) * (W >> sx) samples of its plane, row stride W >> sx
        coeff = np.zeros((3, H, W), np.int32)
        views = [coeff[c].reshape(-1)[:(H >> sy[c]) * (W >> sx[c])].reshape(H >> sy[c], W >> sx[c]) for c in range(3)]
        cols = (W + 255) // 256
        for pass_, grp, q in groups:
            gy, gx = grp // cols, grp % cols
            for c in range(3):
                h, w = q[c].shape
                y0, x0 = (gy * 256) >> sy[c], (gx * 256) >> sx[c]
                views[c][y0:y0 + h, x0:x0 + w] += q[c]
        lfg = []
        for g in lfgroups:
            g = dict(g)
            if g.get("lf_quant") is not None and sub:
                # LFCoefficients.java:66-75 only: chroma-from-luma and smoothing do not apply to subsampled frames
                g["lf"] = [np.ascontiguousarray(g["lf_quant"][c].astype(np.float32) *
                                                (np.float32(g["scaled_dequant"][c]) / np.float32(1 << g["extra_precision"])))
                           for c in range(3)]
            elif g.get("lf_quant") is not None:
                lf = orc.lf_dequant(np.stack(g["lf_quant"]), g["scaled_dequant"], g["extra_precision"], g["x_factor_lf"],
                                    g["b_factor_lf"], g["adaptive_smoothing"], params.base_corr_x, params.base_corr_b,
                                    params.color_factor)
                g["lf"] = [np.ascontiguousarray(lf[c]) for c in range(3)]
            lfg.append(g)
        frame = dict(params=bytes(params), weights=np.ascontiguousarray(weights, np.float32),
                     woffs=np.ascontiguousarray(woffs, np.int32), lfgroups=lfg, coeff=coeff, width=W, height=H)
        self._keep = (frame, lfg)
        return orc.vardct_frame(frame, threads=self.threads)

    def gab(self, planes, w1, w2):
        planes = np.ascontiguousarray(planes, np.float32)
        if planes.shape[0] == 1:  # one colour channel: orc.gab1 / orc.epf1 (Frame.java:642,661: channel 0 in all three rounds)
            return orc.gab1(planes[0], w1[0], w2[0])[None]
        return orc.gab(planes, w1, w2)

    def epf(self, planes, iters, inv_sigma, sigma_modular, rf):
        planes = np.ascontiguousarray(planes, np.float32)
        if planes.shape[0] == 1:
            return orc.epf1(planes[0], iters, inv_sigma, sigma_modular, rf["channel_scale"], rf["pass0"], rf["pass2"], rf["border_sad_mul"])[None]
        return orc.epf(planes, iters, inv_sigma, sigma_modular, rf["channel_scale"], rf["pass0"], rf["pass2"], rf["border_sad_mul"])

    def xyb(self, planes, matrix, opsin_bias, cbrt_bias, intensity_target):
        return orc.xyb(planes, list(matrix), list(opsin_bias), list(cbrt_bias), intensity_target)

    def ycbcr(self, planes):
        return orc.ycbcr(planes)

    def squeeze(self, ins, steps, shapes):
        return orc.modular_apply(ins, steps, rct_type=-1, out_shapes=shapes)

    def rct(self, a, b, c, rct_type):
        return orc.rct(np.stack([a, b, c]), rct_type)

    def modular_to_float(self, a, b, scale):
        return orc.modular_to_float(a, b, scale)

    def chroma_upsample(self, plane, xs, ys):
        return orc.chroma_upsample(plane, xs, ys)

    def upsample(self, plane, k, weights):
        return orc.upsample(plane, k, weights)

    def noise_init(self, h, w, seed0, group_dim, colors):
        return orc.noise_init(h, w, seed0, group_dim, colors)

    def noise_add(self, planes, noise, lut, bcx, bcb):
        return orc.noise_add(planes, noise, lut, bcx, bcb)

    def blend(self, mode, canvas, frame, ref, rect, frameAlpha=None, refAlpha=None, isAlpha=False, hasExtra=False, clamp=False,
              premult=False):
        st, out = orc.blend(mode, canvas, frame, ref, rect, frame_alpha=frameAlpha, ref_alpha=refAlpha, is_alpha=isAlpha,
                            has_extra=hasExtra, clamp=clamp, premult=premult)
        if st != 0:
            raise RuntimeError("oracle blend status %d" % st)
        return out

    def orient(self, plane, orientation):
        return orc.orient(plane, orientation)

    def transfer(self, plane, tf):
        return orc.transfer(plane, {TF_PQ: abi.TRANSFER_PQ, TF_SRGB: abi.TRANSFER_SRGB}[tf], 0)

    def pack(self, planes, bit_depth, alpha, premultiplied, tagged, big_endian):
        return orc.pack(planes, bit_depth, alpha=alpha, premultiplied=premultiplied, tagged_depth=tagged, big_endian=big_endian)
