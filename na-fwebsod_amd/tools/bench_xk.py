import sys, torch
sys.path.insert(0, '/root/repo/na-fwebsod_amd')
from naws_hip import ops
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(1)
r, m, n = 4000, 8192, 25088
x = torch.randn((r, n), device=dev, generator=g).relu_()
dy = torch.randn((r, m), device=dev, generator=g)
dy[torch.rand((r, m), device=dev, generator=g) < 0.75] = 0
xp = ops.split_f16x2(x); del x
a2 = ops.split_f16x2(dy, transpose=True, rowmul=xp.inv_scale)
xt = ops.f16_planes_transpose(xp)
out = torch.empty((m, n), device=dev)
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for rnd in range(3):
    print('nt  24576 cols %.3f ms' % t(lambda: ops.gemm_f32_f16x2_nt(a2, xt.rows(0, 24576), out=out[:, :24576])),
          ' xk 24576 cols %.3f ms' % t(lambda: ops.gemm_f32_f16x2_nt_xk(a2, xp, ncols=(0, 24576), out=out[:, :24576])),
          ' nt 512 %.3f' % t(lambda: ops.gemm_f32_f16x2_nt(a2, xt.rows(24576, n), out=out[:, 24576:])),
          ' xk 512 %.3f' % t(lambda: ops.gemm_f32_f16x2_nt_xk(a2, xp, ncols=(24576, n), out=out[:, 24576:])))
