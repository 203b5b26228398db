"""Experiment harness for the critic kernel: build mlp.hip with -D switches
(sections compiled out; results are wrong, only the time matters) and time each
variant.   build:  python scripts/mlp_variants.py build
           run  :  python scripts/mlp_variants.py run        (on the GPU box)"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "tce_rl_amd", "csrc")
OUT = os.path.join(HERE, "variants")
VARIANTS = {
    "base": [], "stamp": ["-DMLPX_STAMP"],
}
VARIANTS.update({k: v.split() for k, v in
                 (a.split("=", 1) for a in os.environ.get("MLPX_EXTRA", "").split(";") if a)})


def build():
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OUT, exist_ok=True)
    def one(kv):
        name, flags = kv
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared",
               *flags, os.path.join(CSRC, "mlp.hip"), os.path.join(CSRC, "capi.hip"),
               "-o", os.path.join(OUT, "libmlp_%s.so" % name)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        print(name, "ok" if r.returncode == 0 else r.stderr[-2000:], flush=True)
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(one, VARIANTS.items()))


def run():
    import torch
    N, T = 4096, 500
    g = torch.Generator(device="cuda").manual_seed(0)
    full = torch.randn(N, T + 1, 48, device="cuda", generator=g)
    x = full[:, :-1, :40]
    ret = torch.randn(N * T, device="cuda", generator=g)
    din, H = 40, 128
    w1 = torch.randn(H, din, device="cuda") * 0.1; b1 = torch.zeros(H, device="cuda")
    w2 = torch.randn(H, H, device="cuda") * 0.1; b2 = torch.zeros(H, device="cuda")
    w3 = torch.randn(1, H, device="cuda") * 0.1; b3 = torch.zeros(1, device="cuda")
    for name in VARIANTS:
        path = os.path.join(OUT, "libmlp_%s.so" % name)
        if not os.path.exists(path):
            continue
        lib = ctypes.CDLL(path)
        P = lib.tce_mlp_critic_num_params(din); G = lib.tce_mlp_critic_grid()
        partials = torch.empty(G, P + 2, device="cuda"); flat = torch.empty(P, device="cuda"); stats = torch.empty(2, device="cuda")
        vp = ctypes.c_void_p
        fn = lib.tce_mlp_critic_f32
        fn.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int] + [vp] * 6 + \
            [ctypes.c_int, vp, vp, ctypes.c_float, vp, vp, vp, vp, ctypes.c_int, vp, vp, vp, vp] + [ctypes.c_float] * 7 + [vp, vp]
        def go(bwd=True):
            rc = fn(x.data_ptr(), x.stride(0), x.stride(1), T, N * T, din, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                    b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), 1, ret.data_ptr() if bwd else None, None, 0.0,
                    None if bwd else stats.data_ptr() * 0 + partials.data_ptr(), partials.data_ptr() if bwd else None,
                    flat.data_ptr() if bwd else None, stats.data_ptr() if bwd else None, 0, None, None, None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None,
                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0, lib.tce_last_error()
        res = []
        for bwd in (True, False):
            go(bwd); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for rep in range(4):
                s.record()
                for _ in range(10): go(bwd)
                e.record(); torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) / 10)
            res.append(best)
        print(f"{name:10s} fwd+bwd {res[0]:.3f} ms   fwd {res[1]:.3f} ms", flush=True)
        if name.startswith("stamp"):
            go(True); torch.cuda.synchronize()
            st = partials[0, :14].cpu().tolist()
            names = ["forward", "P1a bar+loss+dY2", "barrier P1b", "P2 writes+bar", "B2 loop", "epilogue", "barrier P3", "P4 writes+bar", "-", "loop top",
                     "G: dW1", "G: dW1+barrier", "G: dW2", "G: dW2+barrier"]
            tiles = (N * T // 64) // G
            for n_, v_ in zip(names, st):
                print(f"    {n_:16s} {v_ / tiles:9.0f} cycles/tile (counter units)")


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
