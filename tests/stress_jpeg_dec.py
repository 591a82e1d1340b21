"""GPU: randomized hunt for device-decoder bugs.  Files from the oracle encoder and from Pillow / libjpeg-turbo (optimised tables,
restart intervals of every size) over random sizes, qualities and content; every one decoded on the device (device and host output)
and by the image's libjpeg through the oracle's harness: the planes must be identical.  usage: python tests/stress_jpeg_dec.py [cases] [seed] [damage]; tests/test_gpu_jpeg.py runs 400 cases"""
import ctypes as C
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from libultrahdr_dev_amd import api
from oracle import oracle as orc

def run(cases, seed, dump_dir="gpurun_out", damage=False):
    """-> (identical decodes, mismatches); damage: every file also with bytes of its scan flipped, cut or doubled -- those must
    come back with a status (their planes are not compared: libjpeg conceals what it can, this decoder refuses)"""
    lib = api.init(0)
    rng = np.random.RandomState(seed)


    def content(kind, w, h):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        if kind == 0:
            y = rng.randint(0, 256, (h, w))
        elif kind == 1:
            y = 128 + 100 * np.sin(xx * rng.uniform(0.01, 0.3)) * np.cos(yy * rng.uniform(0.01, 0.3)) + rng.normal(0, rng.uniform(0, 12), (h, w))
        elif kind == 2:
            y = np.full((h, w), rng.randint(0, 256)) + (rng.rand(h, w) < 0.01) * rng.randint(-200, 200)
        else:
            y = (xx + yy) * rng.uniform(0.1, 2.0) + rng.randint(0, 64, (h, w)) * (rng.rand() < 0.5)
        y = np.clip(y, 0, 255).astype(np.uint8)
        c = np.clip(128 + 60 * np.sin(xx[::2, ::2] * 0.05 + rng.uniform(0, 6)) + rng.normal(0, rng.uniform(0, 20), (h // 2, w // 2)), 0, 255).astype(np.uint8)
        c2 = np.clip(128 + 60 * np.cos(yy[::2, ::2] * 0.07 + rng.uniform(0, 6)) + rng.normal(0, rng.uniform(0, 20), (h // 2, w // 2)), 0, 255).astype(np.uint8)
        return np.ascontiguousarray(y), np.ascontiguousarray(c), np.ascontiguousarray(c2)


    def gpu_decode(data, device):
        buf = np.frombuffer(data, np.uint8)
        desc = api.Image()
        cap = 8192 * 64
        probe = lib.uhdr_hip_jpeg_decode(C.c_void_p(buf.ctypes.data), buf.size, None, 0, C.byref(desc), api.MEM_HOST, None)
        need = desc.width * desc.height * (1 if desc.pixelFormat == api.PIX_FMT_MONOCHROME else 3) // (1 if desc.pixelFormat == api.PIX_FMT_MONOCHROME else 2)
        if probe not in (0, api.ERROR_INSUFFICIENT_RESOURCE):
            return probe, None
        if device:
            out = torch.full((need + 64,), 0xCD, dtype=torch.uint8, device="cuda")
            rc = lib.uhdr_hip_jpeg_decode(C.c_void_p(buf.ctypes.data), buf.size, C.c_void_p(out.data_ptr()), need, C.byref(desc), api.MEM_DEVICE, None)
            torch.cuda.synchronize()
            o = out.cpu().numpy()
        else:
            o = np.full(need + 64, 0xCD, np.uint8)
            rc = lib.uhdr_hip_jpeg_decode(C.c_void_p(buf.ctypes.data), buf.size, C.c_void_p(o.ctypes.data), need, C.byref(desc), api.MEM_HOST, None)
        assert (o[need:] == 0xCD).all(), "wrote past the planes"
        return rc, o[:need].copy()


    bad = 0
    decoded = 0
    t0 = time.time()
    for it in range(cases):
        big = rng.rand() < 0.15
        w = 2 * rng.randint(1, 1200 if big else 180)
        h = 2 * rng.randint(1, 700 if big else 130)
        kind = rng.randint(0, 4)
        y, u, v = content(kind, w, h)
        q = int(rng.choice([rng.randint(1, 101), 95, 100, 75, 50]))
        gray = rng.rand() < 0.2
        src = rng.randint(0, 3)
        if src == 0 and w % 2 == 0:
            data = orc.jpeg_encode("orc", y.reshape(-1), None if gray else np.concatenate([u.reshape(-1), v.reshape(-1)]), w, h, q)
            tag = "orc"
        else:
            kw = dict(quality=q)
            if rng.rand() < 0.5:
                kw["optimize"] = True
            r = rng.rand()
            if r < 0.3:
                kw["restart_marker_blocks"] = int(rng.choice([1, 2, 3, 5, 8, 13, 64, 500, 4096]))
            elif r < 0.5:
                kw["restart_marker_rows"] = int(rng.randint(1, 4))
            if gray:
                im = Image.fromarray(y, mode="L")
            else:
                ycc = np.stack([y, np.repeat(np.repeat(u, 2, 0), 2, 1)[:h, :w], np.repeat(np.repeat(v, 2, 0), 2, 1)[:h, :w]], -1)
                im = Image.fromarray(ycc, mode="YCbCr")
                kw["subsampling"] = 2
            b = io.BytesIO()
            im.save(b, "JPEG", **kw)
            data = b.getvalue()
            tag = "pil %s" % kw
        if damage:
            d = bytearray(data)
            sos = d.find(b"\xff\xda")
            for _ in range(rng.randint(1, 6)):
                at = rng.randint(sos + 12, max(sos + 13, len(d) - 2))
                k = rng.randint(0, 4)
                if k == 0: d[at] = rng.randint(0, 256)
                elif k == 1: d[at] = 0xFF
                elif k == 2: del d[at:at + rng.randint(1, 40)]
                else: d[at:at] = bytes(rng.randint(0, 256, rng.randint(1, 9)).astype(np.uint8))
            for device in (True, False):
                rc, got = gpu_decode(bytes(d), device)
                if rc not in (0, api.UNKNOWN_ERROR, api.ERROR_UNSUPPORTED_FEATURE, api.ERROR_RESOLUTION_MISMATCH):
                    bad += 1
                    print("DAMAGED case %d: rc %d" % (it, rc), flush=True)
        st, want, dw, dh, g = orc.jpeg_decode("lj", data)
        for device in (True, False):
            rc, got = gpu_decode(data, device)
            ok = (st > 0 and rc == 0 and np.array_equal(got, want)) or (st <= 0 and rc != 0)
            decoded += 1 if (ok and rc == 0) else 0
            if not ok:
                bad += 1
                print("MISMATCH case %d %dx%d q%d kind %d gray %d %s device %d: lj %d rc %d diff %s" % (
                    it, w, h, q, kind, gray, tag, device, st, rc, "-" if got is None or want is None else int((got != want).sum())), flush=True)
                if bad <= 5:
                    open(os.path.join(dump_dir, "stress_bad_%d.jpg" % it), "wb").write(data)
        if it % 100 == 99:
            print("%d cases, %d mismatches, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)
    print("done: %d cases, %d identical decodes, %d mismatches" % (cases, decoded, bad))
    return decoded, bad


if __name__ == "__main__":
    _, mismatches = run(int(sys.argv[1]) if len(sys.argv) > 1 else 500, int(sys.argv[2]) if len(sys.argv) > 2 else 1, damage="damage" in sys.argv[3:])
    sys.exit(1 if mismatches else 0)
