"""Randomised device-vs-oracle sweep of the Modular path (GPU box): image sizes around the segment / chunk boundaries,
1-4 channels, residual magnitudes from flat to near the int32 limits, RCT types, both forms of the horizontal step, and (r5) the
fused V + H kernel's switches: chunk width 16 / 32, forced segment lengths, fused launches off.
    python tools/fuzz_modular_gpu.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jxlatte_amd import _lib, host, synth
from oracle import pyoracle as orc


def run(n_cases=60, seed=11, device=0, verbose=True):
    """returns the number of mismatching images"""
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(device)
    bad = 0
    sizes = [1, 2, 7, 8, 9, 63, 64, 65, 127, 128, 129, 130, 160, 255, 257, 300, 511, 513, 700, 1023, 1100]
    keys = ("JXL_HSQUEEZE_WALK_MAX", "JXL_VH_CW", "JXL_VH_SEG", "JXL_SQUEEZE_NO_VH")
    saved_all = {k: os.environ.get(k) for k in keys}
    saved = saved_all["JXL_HSQUEEZE_WALK_MAX"]
    try:
        for case in range(n_cases):
            w, h = int(rng.choice(sizes)), int(rng.choice(sizes))
            ch = int(rng.integers(1, 5))
            scale = float(rng.choice([0.0, 1.0, 4.0, 300.0, 2e6, 5e8]))
            os.environ["JXL_HSQUEEZE_WALK_MAX"] = "0" if rng.integers(0, 2) else str(1 << 40)
            for k in ("JXL_VH_CW", "JXL_VH_SEG", "JXL_SQUEEZE_NO_VH"):
                os.environ.pop(k, None)
            sw = int(rng.integers(0, 6))  # 0: the library's choice
            if sw == 1:
                os.environ["JXL_VH_CW"] = "16"
            elif sw == 2:
                os.environ["JXL_VH_CW"] = "32"
            elif sw == 3:
                os.environ["JXL_VH_CW"], os.environ["JXL_VH_SEG"] = "32", str(int(rng.choice([32, 64, 96])))
            elif sw == 4:
                os.environ["JXL_VH_CW"], os.environ["JXL_VH_SEG"] = "16", str(int(rng.choice([16, 48, 80])))
            elif sw == 5:
                os.environ["JXL_SQUEEZE_NO_VH"] = "1"
            mod = synth.make_modular_frame(w, h, channels=ch, seed=int(rng.integers(1, 1 << 30)), res_scale=max(scale, 1e-9))
            if scale == 0.0:
                for a in mod["chans"][ch:]:
                    a[:] = 0
            rct = int(rng.integers(-1, 42)) if ch >= 3 else -1
            ms = host.ModularStream(ctx, mod["chans"], mod["sp"], rctType=rct, rctBegin=0)
            out = ms.applyTransforms()
            exp = orc.modular_apply(mod["chans"], mod["sp"], rct_type=rct, rct_begin=0)
            ok = len(out) == len(exp) and all(np.array_equal(a, b) for a, b in zip(out, exp))
            if not ok:
                bad += 1
                print("MISMATCH case %d: %dx%d ch=%d scale=%g rct=%d walk_max=%s cw=%s seg=%s no_vh=%s" % (
                    case, w, h, ch, scale, rct, os.environ["JXL_HSQUEEZE_WALK_MAX"], os.environ.get("JXL_VH_CW"), os.environ.get("JXL_VH_SEG"), os.environ.get("JXL_SQUEEZE_NO_VH")))
    finally:
        ctx.close()
        for k, v in saved_all.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if verbose:
        print("modular fuzz: %d cases, %d mismatches" % (n_cases, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 11) else 0)
