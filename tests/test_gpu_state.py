"""The library's process-wide state (include/naws.h): the per-(kernel, device) record of raised
dynamic-LDS limits.  SURVEY.md 8(b): entry points re-entrant across streams and devices."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_big_lds_kernels_across_a_launch_state_reset_streams_and_threads(dev):
    """A > 64 KB-LDS kernel (the 256x256 f16x2 GEMM: 128 KB) launched before and after the
    record is dropped, on two streams and from a second host thread: every launch succeeds and
    all results are bit-identical (the attribute call is idempotent and per device)."""
    import threading
    from naws_hip import lib, ops
    g = torch.Generator(device=dev).manual_seed(5)
    a = ops.split_f16x2(torch.randn((512, 1024), device=dev, generator=g))
    b = ops.split_f16x2(torch.randn((512, 1024), device=dev, generator=g))
    ref = ops.gemm_f32_f16x2_nt(a, b).clone()
    assert lib.call('naws_launch_state_reset') == 0
    again = ops.gemm_f32_f16x2_nt(a, b)
    torch.cuda.synchronize()
    assert torch.equal(ref, again)
    outs = []

    def worker():
        torch.cuda.set_device(dev)
        st = torch.cuda.Stream(device=dev)
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            lib.call('naws_launch_state_reset')
            outs.append(ops.gemm_f32_f16x2_nt(a, b))
        st.synchronize()
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert len(outs) == 1 and torch.equal(ref, outs[0])


def test_variant_knobs_do_not_change_results(dev):
    from naws_hip import lib, ops
    g = torch.Generator(device=dev).manual_seed(6)
    a = ops.split_f16x2(torch.randn((384, 512), device=dev, generator=g))
    b = ops.split_f16x2(torch.randn((640, 512), device=dev, generator=g))
    ref = ops.gemm_f32_f16x2_nt(a, b).clone()
    try:
        for v in (5, 17):                  # forms present in the product library (K <= 1024)
            lib.set_variant('h2', v)
            out = ops.gemm_f32_f16x2_nt(a, b)
            # another tile form may accumulate K in another grouping: fp32-accumulation close
            assert torch.allclose(out, ref, rtol=1e-5, atol=1e-4 * float(ref.abs().max()))
    finally:
        lib.set_variant('h2', 0)
