"""Dev probe: the observation-store pattern of a recording rollout as a micro-benchmark (scripts/micro/obs_store_pattern.hip,
built to build_probe/libobs_pattern.so) next to a plain fill of the same bytes."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from probe_records import timed

if __name__ == "__main__":
    lib = ctypes.CDLL(os.path.join(root, "build_probe", "libobs_pattern.so"))
    lib.obs_pattern.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    for b in (65536,):
        T = 20
        out = torch.empty(T * b * 60, device="cuda")
        mb = out.numel() * 4 / 1e6
        def run(v, spin=0, steps=T):
            def f():
                for _ in range(5):
                    rc = lib.obs_pattern(v, out.data_ptr(), b, steps, torch.cuda.current_stream().cuda_stream, spin)
                    assert rc == 0
            return timed(f) / 5 * 1e6
        def fill():
            for _ in range(5): out.fill_(1.0)
        tf = timed(fill) / 5 * 1e6
        line = [f"fill {tf:7.1f} us ({mb / tf / 1e3 * 1e3:.0f} GB/s)"]
        for v, name in ((0, "stores only"), (1, "through LDS"), (2, "half slabs, 2x waves")):
            t = run(v)
            line.append(f"{name} {t:7.1f} us ({mb / t:.0f} GB/s)".replace("GB/s", "MB/us"))
        print(f"b={b} {mb:.0f} MB: " + "  ".join(line), flush=True)
        for spin in (0, 30, 60, 120, 250):
            t0 = run(3, spin, 0) if False else None
            print(f"  spin {spin}: stores+spin {run(3, spin):7.1f} us   spread {run(4, spin):7.1f} us   staggered {run(5, spin):7.1f} us   spin only {run(6, spin):7.1f} us   nt stores+spin {run(7, spin):7.1f} us", flush=True)
