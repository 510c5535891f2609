"""Is a 3-way bf16 split of fp32 factors (6 bf16 MFMA products, fp32 accumulate) as accurate as the fp32 MFMA path?
Feasibility check with the EXISTING bf16 fused kernel on K-expanded operands (r = 32 -> K' = 192)."""
import sys
import torch
sys.path.insert(0, '.')
from teamoflow_amd import _ops

torch.manual_seed(0)
dev = 'cuda'
m, n, r, k = 8192, 16384, 32, 10


def split3(x):
    a1 = x.to(torch.bfloat16)
    r1 = x - a1.float()
    a2 = r1.to(torch.bfloat16)
    r2 = r1 - a2.float()
    a3 = r2.to(torch.bfloat16)
    return a1, a2, a3, (r2 - a3.float()).abs().max().item()


for scale in (1.0, 1e-4):
    U = (torch.randn(m, r, device=dev) * scale).float()
    V = (torch.randn(n, r, device=dev) * scale).float()
    ref = (U.double() @ V.double().T)
    rv, ri = torch.topk(ref, k, dim=1)
    norm = ref.abs().max().item()
    v32, i32 = _ops.predict_topk(U, V, k, return_values=True)
    e32 = (v32.double() - torch.gather(ref, 1, i32.long())).abs().max().item() / norm
    a1, a2, a3, resa = split3(U)
    b1, b2, b3, resb = split3(V)
    orders = {
        'big_first': ([a1, a1, a2, a2, a1, a3], [b1, b2, b1, b2, b3, b1]),
        'small_first': ([a3, a1, a2, a2, a1, a1], [b1, b3, b2, b1, b2, b1]),
        'three_products': ([a2, a1, a1], [b1, b2, b1]),
    }
    print(f'scale {scale}: split residuals {resa:.3g} {resb:.3g}; fp32 MFMA kernel: max |err| / max|ref| = {e32:.3g}, '
          f'index rows equal to fp64 top-k: {(i32.long() == ri).all(1).float().mean().item():.4f}')
    for name, (aa, bb) in orders.items():
        A = torch.cat(aa, 1).contiguous()
        B = torch.cat(bb, 1).contiguous()
        vv, ix = _ops.predict_topk(A, B, k, return_values=True)
        err = (vv.double() - torch.gather(ref, 1, ix.long())).abs().max().item() / norm
        rel_el = ((vv.double() - torch.gather(ref, 1, ix.long())).abs() / torch.gather(ref, 1, ix.long()).abs()).max().item()
        print(f'   {name:15s} max |err| / max|ref| = {err:.3g}   max elementwise rel = {rel_el:.3g}   rows equal to fp64 top-k: '
              f'{(ix.long() == ri).all(1).float().mean().item():.4f}   equal to fp32 kernel: {(ix == i32).all(1).float().mean().item():.4f}')
