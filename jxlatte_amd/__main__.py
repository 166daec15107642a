"""python -m jxlatte_amd in.jxl [out.png] -- the reference's CLI (J/JXLatte.java) on the MI355X back-end:
decodes a JPEG XL file (C++ front-end -> device library) and writes a PNG (HDR images as 16-bit BT.2100 PQ, like
--png-hdr=auto). Needs a GPU: there is no CPU fallback."""
import argparse
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m jxlatte_amd", description=__doc__)
    ap.add_argument("input")
    ap.add_argument("output", nargs="?")
    ap.add_argument("--png-hdr", choices=["auto", "yes", "no"], default="auto")
    ap.add_argument("--png-depth", type=int, default=-1)
    ap.add_argument("--info", action="store_true", help="print the image / frame headers and stop")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    from . import frontend
    if a.info:
        fe = frontend.Frontend(open(a.input, "rb").read())
        im = fe.image
        print("Image: %s\n    Size: %dx%d\n    Bit Depth: %d\n    Extra Channels: %d\n    XYB Encoded: %s\n    Orientation: %d" % (
            a.input, im.width, im.height, im.bits_per_sample, im.num_extra, bool(im.xyb_encoded), im.orientation))
        return 0
    from .decoder import DeviceBackend, JXLDecoder, PNGWriter
    t0 = time.time()
    backend = DeviceBackend(a.device)
    dec = JXLDecoder(a.input, backend=backend)
    image = dec.decode()
    if image is None:
        print("jxlatte_amd: no frames", file=sys.stderr)
        return 1
    t1 = time.time()
    for i, st in enumerate(dec.stats):
        print("    frame %d: %s %dx%d, %d groups" % (i, st["encoding"], st["width"], st["height"], st["groups"]), file=sys.stderr)
    print("Decoded %dx%d in %.3f s" % (image.getWidth(), image.getHeight(), t1 - t0), file=sys.stderr)
    if a.output:
        hdr = image.isHDR() if a.png_hdr == "auto" else a.png_hdr == "yes"
        with open(a.output, "wb") as f:
            PNGWriter(image, bitDepth=16 if hdr else a.png_depth, hdr=hdr).write(f)
    backend.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
