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

# what triggers it: rows, elements or bytes?  float rows of 256 and 64 elements, int64 rows of 2, row-gather by an index list
for shape, dt in (((1_000_000, 256), torch.float32), ((1_100_000, 256), torch.float32), ((2_000_000, 256), torch.float32), ((4_200_000, 256), torch.float32),
                  ((8_000_000, 64), torch.float32), ((20_000_000, 64), torch.float32), ((60_000_000, 2), torch.int64), ((70_000_000, 2), torch.int64),
                  ((140_000_000, 2), torch.int32), ((280_000_000, 2), torch.int32)):
    n = shape[0]
    x = (torch.arange(n, device=dev, dtype=torch.int64)[:, None] + torch.arange(shape[1], device=dev)[None, :]).to(dt)
    sel = torch.arange(0, n, 3, device=dev)
    a = x[sel]
    ok = bool(torch.equal(a[:, 0].to(torch.int64), sel)) and bool(torch.equal(a[:, -1].to(torch.int64), sel + shape[1] - 1))
    print(shape, str(dt), 'bytes %.2f GiB' % (x.numel() * x.element_size() / 2 ** 30), 'x[index] correct:', ok, flush=True)
    del x, a, sel
