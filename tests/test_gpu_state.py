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


def test_background_streams_priority_and_cu_mask(dev):
    """naws_stream_create: a low-priority stream and one confined to every fourth compute unit
    run the same kernel to the same bits; argument errors come back as codes."""
    import ctypes as C
    from naws_hip import lib, ops
    g = torch.Generator(device=dev).manual_seed(7)
    a = ops.split_f16x2(torch.randn((512, 1024), device=dev, generator=g))
    b = ops.split_f16x2(torch.randn((512, 1024), device=dev, generator=g))
    ref = ops.gemm_f32_f16x2_nt(a, b).clone()
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    for kw in (dict(priority=1), dict(priority=-1), dict(cu_mask=ops.cu_mask_every(n_cu, 4))):
        bs = ops.BackgroundStream(dev, **kw)
        bs.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(bs.stream):
            out = ops.gemm_f32_f16x2_nt(a, b)
        bs.stream.synchronize()
        assert torch.equal(out, ref), kw
        bs.close()
    L = lib.load()
    h = C.c_void_p()
    assert L.naws_stream_create(0, None, 0, None) == lib.ERR_NULL
    assert L.naws_stream_create(0, None, 2, C.byref(h)) == lib.ERR_ARG
    assert L.naws_stream_destroy(None) == lib.OK


def test_engine_update_on_a_masked_stream_is_the_same_update(dev):
    """The stream the deferred SGD runs on only moves it to another queue (the engine uses an
    ordinary torch stream; a masked or low-priority one from naws_stream_create is injected
    here): parameters after three steps are bit-identical."""
    import numpy as np
    from detectron.datasets import synthetic
    from naws_hip import ops
    from naws_hip.engine import WsddnEngine
    c, B = 20, 2
    mb = synthetic.make_minibatch(synthetic.make_roidb(B, 64, c, 96, 128, seed=3), c)
    t = {k: torch.from_numpy(v).to(dev) for k, v in mb.items()}
    seg = [0] + np.cumsum(np.bincount(mb['rois'][:, 0].astype(np.int64), minlength=B)).tolist()
    blobs = synthetic.init_blobs(c, seed=3)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    res, keep = [], []
    for kw in (None, dict(cu_mask=ops.cu_mask_every(n_cu, 4)), dict(priority=1)):
        eng = WsddnEngine(c + 1, dev, gpu_num=B, seed=3)
        eng.set_conv_blobs(blobs)
        eng.set_head_blobs(blobs)
        eng.set_lr(1e-4)
        if kw is not None:
            keep.append(ops.BackgroundStream(dev, **kw))
            eng._upd_stream = keep[-1].stream
        for _ in range(3):
            eng.forward_backward(t['data'], t['rois'], t['obn_scores'], t['labels_oh'], seg=seg)
            eng.sgd_step()
        eng.flush()
        torch.cuda.synchronize()
        res.append(eng.params.clone())
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])
