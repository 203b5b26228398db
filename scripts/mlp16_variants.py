"""Experiment harness for the split-f16 (mlp16.hip) and three-part bf16
(mlpb.hip: set M16_KERNEL=mlpb) critic kernels: build the file with -D switches
and time each variant; the `stamp` variant prints cycles per phase.
   build:  python scripts/mlp16_variants.py build
   run  :  python scripts/mlp16_variants.py run        (on the GPU box)
   extra variants: M16_EXTRA="name=-DFLAG ...;name2=..." """
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "tce_rl_amd", "csrc")
OUT = os.path.join(HERE, "variants")
KERNEL = os.environ.get("M16_KERNEL", "mlp16")
ENTRY = {"mlp16": "tce_mlp_critic_f16x2", "mlpb": "tce_mlp_critic_bf16x3"}[KERNEL]
VARIANTS = {"base": [], "stamp": ["-DM16_STAMP", "-DMLPB_STAMP"]}
VARIANTS.update({k: v.split() for k, v in
                 (a.split("=", 1) for a in os.environ.get("M16_EXTRA", "").split(";") if a)})


def build():
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OUT, exist_ok=True)
    def one(kv):
        name, flags = kv
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared",
               *flags, os.path.join(CSRC, KERNEL + ".hip"), os.path.join(CSRC, "capi.hip"),
               os.path.join(CSRC, "xchg.hip"), os.path.join(CSRC, "optim.hip"),
               "-o", os.path.join(OUT, "lib%s_%s.so" % (KERNEL, name))]
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
        path = os.path.join(OUT, "lib%s_%s.so" % (KERNEL, name))
        if not os.path.exists(path):
            continue
        lib = ctypes.CDLL(path)
        P = H * din + H + H * H + H + H + 1; G = 256
        partials = torch.empty(G, P + 2, device="cuda"); flat = torch.empty(P, device="cuda"); stats = torch.empty(2, device="cuda")
        vp = ctypes.c_void_p
        fn = getattr(lib, ENTRY)
        fn.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int] + [vp] * 6 + \
            [ctypes.c_int, vp, vp, ctypes.c_float, vp, vp, vp, vp, ctypes.c_int, vp, vp, vp, vp] + [ctypes.c_float] * 7 + [vp, vp]
        def go():
            rc = fn(x.data_ptr(), x.stride(0), x.stride(1), T, N * T, din, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                    b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), 1, ret.data_ptr(), None, 0.0, None, partials.data_ptr(),
                    flat.data_ptr(), stats.data_ptr(), 0, None, None, None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None,
                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        go(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(2_000_000); s.record()
            for _ in range(10): go()
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 10)
        print(f"{name:12s} {best*1e3:8.1f} us/epoch", flush=True)
        if "_STAMP" in " ".join(VARIANTS[name]) or name == "stamp":
            go(); torch.cuda.synchronize()
            st = partials[0, :16].cpu().tolist()
            names = ["x split", "F2", "F4", "loss+dY2+pack", "barrier P1", "P2 writes", "barrier P2", "P3 dH1", "barrier P3",
                     "P4 writes", "barrier P4", "-", "G: dW1", "G: barriers 1+2", "G: dW2", "G: barriers 3+4"]
            if KERNEL == "mlpb":
                names = ["L1 + H1 image", "barrier A", "L2 + v partials", "barrier B",
                         "loss + dY2 image", "barrier C", "dW2", "dH1 + dY1", "barrier D", "dY1 image + dW1", "barrier E", "X image", "barrier F", "dW1: row loads issued", "dW1: dY1 rows 0-31 written", "tile head"]
            tiles = (N * T // 64) // G
            for n_, v_ in zip(names, st):
                print(f"    {n_:16s} {v_ / tiles:9.0f} counter units/tile")


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
