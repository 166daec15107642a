#!/usr/bin/env python3
"""Generate the committed golden fixtures (tests/golden/*.npz).

The reference is Java and cannot run here or on the GPU box (no JVM), and it ships no vectors of its
own, so these fixtures are produced by the repo's CPU oracle (oracle/jxl_oracle.c, a line-by-line
restatement of the reference). They pin today's oracle outputs (regression guard for the oracle AND
the expected values for the HIP path); the oracle itself is anchored by tests/test_oracle_kats.py.
Every fixture stores ALL inputs (including the quant weights actually used) and the expected outputs.

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jxlatte_amd import abi, synth  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MIX_SMALL = {n: 1.0 for n in ("DCT8", "HORNUSS", "DCT2", "DCT4", "DCT16", "DCT32", "DCT16_8", "DCT8_16", "DCT32_8", "DCT8_32",
                              "DCT32_16", "DCT16_32", "DCT4_8", "DCT8_4", "AFV0", "AFV1", "AFV2", "AFV3", "DCT64")}


def frame_to_npz(fr):
    d = dict(width=fr["width"], height=fr["height"], params=np.frombuffer(bytes(fr["params"]), np.uint8).copy(),
             coeff=fr["coeff"], woffs=fr["woffs"])
    w = fr["weights"].copy()
    w[fr["woffs"][13 * 3]:] = 1.0  # parameter sets 13..16 (128/256-edge blocks) are unused here: keep the file small
    d["weights"] = w
    g = fr["lfgroups"][0]
    assert len(fr["lfgroups"]) == 1
    for k in ("dct_select", "hf_mul", "sharpness", "x_from_y", "b_from_y", "block_yx"):
        d["g_" + k] = g[k]
    d["g_lf"] = np.stack(g["lf"])
    return d


def npz_to_frame(z):
    raw = z["params"].tobytes()  # fixtures written before a struct extension: new trailing fields default to zero
    p = abi.VarDCTParams.from_buffer_copy(raw + bytes(max(0, C.sizeof(abi.VarDCTParams) - len(raw))))
    g = dict(lfg_y=0, lfg_x=0, dct_select=np.ascontiguousarray(z["g_dct_select"]), hf_mul=np.ascontiguousarray(z["g_hf_mul"]),
             sharpness=np.ascontiguousarray(z["g_sharpness"]), x_from_y=np.ascontiguousarray(z["g_x_from_y"]),
             b_from_y=np.ascontiguousarray(z["g_b_from_y"]), block_yx=np.ascontiguousarray(z["g_block_yx"]),
             lf=[np.ascontiguousarray(z["g_lf"][c]) for c in range(3)])
    return dict(params=p, weights=np.ascontiguousarray(z["weights"]), woffs=np.ascontiguousarray(z["woffs"]), lfgroups=[g],
                coeff=np.ascontiguousarray(z["coeff"]), width=int(z["width"]), height=int(z["height"]))


def post_inputs():
    """inputs of the rows f4 / f3 fixture (finite values only, so every expected bit is platform-independent)"""
    rng = np.random.default_rng(4242)
    d = dict(plane=(rng.standard_normal((20, 28)) * 0.3 + 0.4).astype(np.float32),
             up_packed2=(rng.standard_normal(15) * 0.2).astype(np.float32),
             up_packed4=(rng.standard_normal(55) * 0.2).astype(np.float32),
             xyb=(rng.random((3, 20, 28)) * 0.8).astype(np.float32), noise_lut=(rng.random(8) * 0.9).astype(np.float32),
             frame=rng.random((16, 30)).astype(np.float32), frame_alpha=(rng.random((16, 30)) * 1.3 - 0.1).astype(np.float32),
             ref=rng.random((20, 28)).astype(np.float32), ref_alpha=(rng.random((20, 28)) * 0.9 + 0.05).astype(np.float32),
             ints=rng.integers(-50, 5000, (3, 20, 28)).astype(np.int32))
    return d


POST_RECT = (12, 20, 4, 5, 2, 6, 4, 5)
POST_SEED = (5 << 32) | 11


def post_vectors(d=None):
    d = dict(d or post_inputs())
    d["chroma_11"] = orc.chroma_upsample(d["plane"], 1, 1)
    d["chroma_20"] = orc.chroma_upsample(d["plane"], 2, 0)
    for k in (2, 4):
        w = orc.upsampling_weights(k, d["up_packed%d" % k])
        d["up%d" % k] = orc.upsample(d["plane"], k, w)
    nz = orc.noise_init(20, 28, POST_SEED, group_dim=16)
    d["noise"] = nz
    d["noise_added"] = orc.noise_add(d["xyb"], nz, d["noise_lut"], 0.0, 1.0)
    for mode, kw in ((abi.BLEND_ADD, {}), (abi.BLEND_MULT, dict(clamp=True)), (abi.BLEND_BLEND, dict(has_extra=True, clamp=True)),
                     (abi.BLEND_BLEND, dict(has_extra=True, premult=True)), (abi.BLEND_MULADD, dict(has_extra=True))):
        st, out = orc.blend(mode, d["ref"], d["frame"], d["ref"], POST_RECT, frame_alpha=d["frame_alpha"], ref_alpha=d["ref_alpha"], **kw)
        assert st == 0
        d["blend_%d_%d" % (mode, 1 if kw.get("premult") else 0)] = out
    d["orient6"] = orc.orient(d["ints"][0], 6)
    d["orient7"] = orc.orient(d["plane"], 7)
    d["pack_rgb8"] = orc.pack(list(d["xyb"]), 8)
    d["pack_rgba16be"] = orc.pack(list(d["ints"]), 16, alpha=d["ref_alpha"], premultiplied=True, tagged_depth=[12, 12, 12, 8], big_endian=True)
    return d


def main():
    # 1. VarDCT frame, every varblock type up to 64x64, aligned and unaligned tilings, all stage prefixes
    for name, aligned, seed in (("vardct_aligned", True, 2024), ("vardct_unaligned", False, 2025)):
        fr = synth.make_vardct_frame(128, 64, seed=seed, mix=MIX_SMALL, aligned=aligned)
        d = frame_to_npz(fr)
        fr2 = npz_to_frame(d)
        for tag, st in ((("idct", 1), ("gab", 3), ("epf", 7), ("xyb", 15)) if aligned else (("idct", 1), ("xyb", 15))):
            d["expect_" + tag] = orc.vardct_frame(fr2, stages=st)
        # quantised sRGB 8-bit output (ints are exact even if pow differs by an ulp somewhere: checked <= 1 LSB)
        fr2["params"].transfer, fr2["params"].out_format = abi.TRANSFER_SRGB, abi.OUT_U8
        d["expect_srgb_u8"] = orc.vardct_frame(fr2, stages=31).astype(np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    # 2. EPF with 3 iterations + no Gab on a ragged frame
    fr = synth.make_vardct_frame(72, 40, seed=77, mix={"DCT8": 0.6, "DCT16": 0.3, "AFV2": 0.1}, epf_iters=3, gab=False)
    d = frame_to_npz(fr)
    d["expect_xyb"] = orc.vardct_frame(npz_to_frame(d), stages=15)
    np.savez_compressed(os.path.join(HERE, "vardct_epf3_nogab.npz"), **d)
    # 3. Modular: squeezed residual channels -> image (default plan) and RCT
    mod = synth.make_modular_frame(53, 37, channels=3, seed=11)
    out = orc.modular_apply(mod["chans"], mod["sp"], rct_type=13, rct_begin=0)
    np.savez_compressed(os.path.join(HERE, "modular_53x37.npz"), sp=np.array(mod["sp"], np.int32), rct_type=13,
                        **{"chan%d" % i: c for i, c in enumerate(mod["chans"])}, **{"out%d" % i: c for i, c in enumerate(out)})
    # 4. stage vectors
    rng = np.random.default_rng(99)
    planes = (rng.standard_normal((3, 24, 40)) * 0.2).astype(np.float32)
    sig = (rng.random((3, 5)) * 4).astype(np.float32)
    sig[0, 0] = np.inf
    st = dict(planes=planes, inv_sigma=sig,
              gab=orc.gab(planes, [0.115169525] * 3, [0.061248592] * 3),
              epf1=orc.epf(planes, 1, sig, 0.0, (40.0, 5.0, 3.5), 0.9, 6.5, 2.0 / 3.0),
              epf2=orc.epf(planes, 2, sig, 0.0, (40.0, 5.0, 3.5), 0.9, 6.5, 2.0 / 3.0),
              epf3=orc.epf(planes, 3, sig, 0.0, (40.0, 5.0, 3.5), 0.9, 6.5, 2.0 / 3.0))
    p = synth.default_params(8, 8)
    st["xyb"] = orc.xyb(planes, list(p.opsin_matrix), list(p.opsin_bias), list(p.cbrt_opsin_bias), 255.0)
    x = rng.standard_normal((32, 64)).astype(np.float32)
    st["dct_in"] = x
    st["idct_32x64"] = orc.idct2d(x)
    st["idct_32x64_t"] = orc.idct2d(x, transposed=True)
    st["fdct_32x64"] = orc.fdct2d(x)
    np.savez_compressed(os.path.join(HERE, "stages.npz"), **st)
    np.savez_compressed(os.path.join(HERE, "post.npz"), **post_vectors())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
