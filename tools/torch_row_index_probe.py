import torch
dev = torch.device('cuda')
for n in (10_000_000, 70_000_000, 100_000_000, 140_000_000):
    idx = torch.stack([torch.arange(n, device=dev) // 100, torch.arange(n, device=dev) % 100000], 1)
    keep = idx[:, 0] >= 0
    a = idx[keep]
    ok1 = bool(torch.equal(a, idx))
    sel = torch.nonzero(keep).flatten()
    b = idx.index_select(0, sel)
    ok2 = bool(torch.equal(b, idx))
    c = torch.stack([idx[:, 0][keep], idx[:, 1][keep]], 1)
    ok3 = bool(torch.equal(c, idx))
    print(n, 'mask-index', ok1, 'index_select', ok2, 'columnwise', ok3, 'max', int(a.max()), flush=True)
    del idx, keep, a, b, c, sel

# the combined form used by metric code: idx[mask, 0]
for n in (10_000_000, 100_000_000):
    idx = torch.stack([torch.arange(n, device=dev) // 100, torch.arange(n, device=dev) % 100000], 1)
    keep = idx[:, 1] % 3 != 0
    a = idx[keep, 0]
    b = idx[:, 0][keep]
    print(n, 'idx[mask, 0] == idx[:, 0][mask]:', bool(torch.equal(a, b)), flush=True)
    del idx, keep, a, b
