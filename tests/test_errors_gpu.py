"""Error behaviour of the boundary: the product path refuses CPU tensors (no
fallback), the C ABI validates its arguments and reports through
tce_last_error (RuntimeError on the Python side), limits are enforced, and a
launch on bad arguments never reaches the device."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cpu_tensors_are_refused():
    from tce_rl_amd import ops
    r, v = torch.randn(4, 8), torch.randn(4, 9)
    d = torch.zeros(4, 8, dtype=torch.bool)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gae(r, v, d, d, 1.0, 0.95)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.moments(r)
    from tce_rl_amd.nn import MLP
    mlp = MLP("ValueFunction", 4, 1, [8, 8], "orthogonal", 1.0, "relu", None,
              torch.float32, torch.device("cuda"))
    with pytest.raises(RuntimeError):
        mlp(torch.randn(3, 4))


def test_c_abi_reports_bad_arguments():
    from tce_rl_amd import _lib
    from tce_rl_amd._lib import call, ptr, stream
    lib = _lib.load()
    x = torch.randn(10, device="cuda")
    # null buffers / bad sizes -> non-zero return code + message, no launch
    rc = lib.tce_gae_f32(None, None, None, None, None, None, None, 0, None,
                         None, 4, 8, 1.0, 0.95, 1, None)
    assert rc != 0 and b"gae" in lib.tce_last_error()
    with pytest.raises(RuntimeError, match="K <= 64"):
        call("tce_kl_cov_part_f32", 0, ptr(x), ptr(x), 0, None, ptr(x), None,
             1, 65, stream())
    with pytest.raises(RuntimeError, match="D_in"):
        call("tce_mlp_critic_f32", ptr(x), 0, 41, 1, 1, 41, *([ptr(x)] * 6), 1,
             None, None, 0.0, ptr(x), None, None, None, 0, None, None, None,
             None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None, stream())
    with pytest.raises(RuntimeError, match="D_in"):
        call("tce_mlp_critic_f16x2", ptr(x), 0, 41, 1, 1, 41, *([ptr(x)] * 6),
             1, ptr(x), None, 0.0, None, ptr(x), ptr(x), ptr(x), 0, None, None,
             None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None, stream())
    with pytest.raises(RuntimeError, match="backward buffers"):
        call("tce_mlp_critic_f16x2", ptr(x), 0, 8, 1, 1, 8, *([ptr(x)] * 6), 1,
             None, None, 0.0, ptr(x), None, None, None, 0, None, None, None,
             None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None, stream())
    with pytest.raises(RuntimeError, match="D_in"):
        call("tce_mlp_critic_bf16x3", ptr(x), 0, 41, 1, 1, 41, *([ptr(x)] * 6),
             1, ptr(x), None, 0.0, None, ptr(x), ptr(x), ptr(x), 0, None, None,
             None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None, stream())
    with pytest.raises(RuntimeError, match="backward buffers"):
        call("tce_mlp_critic_bf16x3", ptr(x), 0, 8, 1, 1, 8, *([ptr(x)] * 6), 1,
             None, None, 0.0, ptr(x), None, None, None, 0, None, None, None,
             None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None, stream())
    with pytest.raises(RuntimeError, match="CU range"):
        import ctypes
        h = ctypes.c_void_p()
        call("tce_stream_create_cu_range", 31, 2, ctypes.byref(h))
    torch.cuda.synchronize()                 # the device is still healthy
    assert torch.isfinite(x).all()


def test_limits_fall_back_or_raise_cleanly():
    from tce_rl_amd import critic_ops, mlp_ops, ops
    from tce_rl_amd.nn import MLP
    mlp_ops.LIBRARY_CALLS.clear()
    wide = MLP("ValueFunction", 41, 1, [128, 128], "orthogonal", 1.0, "relu",
               None, torch.float32, torch.device("cuda"))
    assert not critic_ops.supported(wide)                  # D_in > 40
    y = wide(torch.randn(5000, 41, device="cuda"))         # generic dense layers
    assert y.shape == (5000, 1) and torch.isfinite(y).all()
    f64 = MLP("ValueFunction", 20, 1, [128, 128], "orthogonal", 1.0, "relu",
              None, torch.float64, torch.device("cuda"))
    assert critic_ops.supported(f64) and not critic_ops.narrow_supported(f64)
    odd = MLP("ValueFunction", 20, 1, [192, 192], "orthogonal", 1.0, "relu",
              None, torch.float32, torch.device("cuda"))
    assert not critic_ops.supported(odd)                   # widths 128 / 256 only
    y = odd(torch.randn(5000, 20, device="cuda"))          # generic dense layers
    assert y.shape == (5000, 1) and torch.isfinite(y).all()
    # (shapes outside the fused families stay on
    # hand-written kernels -- csrc/glin.hip -- under autograd)
    assert not mlp_ops.LIBRARY_CALLS, dict(mlp_ops.LIBRARY_CALLS)
    big64 = MLP("ValueFunction", 40, 1, [256, 256], "orthogonal", 1.0, "relu",
                None, torch.float64, torch.device("cuda"))
    assert not critic_ops.supported(big64)                 # fp64 x 256: D_in <= 24
    K = 65                                                  # K <= 64 everywhere
    L = torch.eye(K, device="cuda").expand(3, K, K).contiguous()
    with pytest.raises(RuntimeError):
        ops.kl_cov_projection(L, L, 1e-3)
    with pytest.raises(NotImplementedError):
        ops.moments(torch.zeros(4, device="cuda", dtype=torch.float16))


def test_missing_library_fails_loudly(monkeypatch):
    from tce_rl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libtce_hip.so")
    with pytest.raises(ImportError, match="no CPU"):
        _lib.load()
