import sys, torch
sys.path.insert(0, '/root/repo/na-fwebsod_amd')
from naws_hip import ops, lib as L
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(1)
R, H = 4000, 4096
dz7 = torch.randn((2, R, H), device=dev, generator=g)
dz7[torch.rand((2, R, H), device=dev, generator=g) < 0.75] = 0
w7 = torch.randn((2, H, H), device=dev, generator=g) * 0.01
h6 = torch.randn((2, R, H), device=dev, generator=g).relu_()
a = ops.split_f16x2(dz7)
bt = ops.split_f16x2(w7, transpose=True)
out = torch.empty((2, R, H), device=dev)
def t(fn, it=8):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
sc = ops.amax_scales(2, R, dev); sct = ops.amax_scales(2, H, dev)
for rnd in range(2):
    print('plain %.3f' % t(lambda: ops.gemm_f32_f16x2_nt(a, bt, out=out)),
          ' gate+aux %.3f' % t(lambda: ops.gemm_f32_f16x2_nt(a, bt, out=out, epilogue=L.EPI_GATE_POS, aux=h6, alpha=2.0)),
          ' maxima %.3f' % t(lambda: ops.gemm_f32_f16x2_nt(a, bt, out=out, rowmax=ops.amax_words(sc), colmax=ops.amax_words(sct))),
          ' both %.3f' % t(lambda: ops.gemm_f32_f16x2_nt(a, bt, out=out, epilogue=L.EPI_GATE_POS, aux=h6, alpha=2.0, rowmax=ops.amax_words(sc), colmax=ops.amax_words(sct))))
