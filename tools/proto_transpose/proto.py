"""Times the weight-transposition prototype (proto.hip) at the C4 shape: 1M users x 1024 negatives -> the (user block, item, user)
entry order of the item pass, tiles of (6,135 users x ~480 items) through LDS.  Checks the permutation on the real index structure
(destination = entries sorted by (tile, item, user); the positives - 9 % of the entries - are left out: they would add a second run
per (user, tile) of the same kind)."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from teamoflow_amd.mf.utils import random_sampler_device

dev = torch.device('cuda', 0)
m, n, S = int(os.environ.get('M', 1_000_000)), 100_000, 1024
C = int(os.environ.get('BLOCKS', 163)); n_jt = int(os.environ.get('JTILES', 209))
upb = -(-m // C); width = -(-n // n_jt)
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libproto_t.so'))
R = torch.sort(random_sampler_device(n, m, S, seed=5, device=dev).to(torch.int64), dim=1)[0]      # item-sorted negatives, as the sliced pass keeps them
D = torch.rand(m, S, device=dev)
# lo[jt][u] = first negative of user u with item >= jt * width
bnd = (torch.arange(n_jt + 1, device=dev) * width)[None, :].expand(m, -1).contiguous()
lo = torch.searchsorted(R, bnd).to(torch.int32)                                                    # [m, n_jt + 1]
cnt = lo[:, 1:] - lo[:, :-1]                                                                       # entries of (u, jt)
blk = torch.arange(m, device=dev) // upb
# source base of (u, jt) inside its tile: exclusive cumsum of cnt over the users of the block
cs = torch.cumsum(cnt.to(torch.int64), 0)
first = torch.arange(C, device=dev) * upb
base0 = torch.cat([torch.zeros(1, n_jt, dtype=torch.int64, device=dev), cs[first[1:] - 1]])        # [C, n_jt]: sum before the block
sb = cs - cnt - base0[blk]                                                                         # [m, n_jt]
tile_cnt = torch.zeros(C, n_jt, dtype=torch.int64, device=dev).index_add_(0, blk, cnt.to(torch.int64))
assert int(tile_cnt.max()) < 65536 and int(sb.max()) < 65536, (int(tile_cnt.max()), int(sb.max()))
lo_sb = torch.zeros(n_jt + 1, m, dtype=torch.int64, device=dev)
lo_sb[:, :] = lo.t().to(torch.int64)
lo_sb[:n_jt] |= (sb.t() << 16)
lo_sb = torch.where(lo_sb >= 2 ** 31, lo_sb - 2 ** 32, lo_sb).to(torch.int32)                          # the bits of a uint32
lo_sb = lo_sb.contiguous()
tile_ptr = torch.zeros(C * n_jt + 1, dtype=torch.int64, device=dev)
tile_ptr[1:] = torch.cumsum(tile_cnt.reshape(-1), 0)
E = int(tile_ptr[-1]); assert E == m * S
# destination order: (block, jt, item, user); source-local position of every entry
u_of = torch.arange(m, device=dev).repeat_interleave(S)
pos = torch.arange(S, device=dev).repeat(m)
item = R.reshape(-1)
jt_of = item // width
src_pos = sb[u_of, jt_of] + (pos - lo[u_of, jt_of].to(torch.int64))                                # position in the tile's source order
key = ((blk[u_of] * n_jt + jt_of) * (width + 1) + (item - jt_of * width)) * upb + (u_of - blk[u_of] * upb)
del item, jt_of
order = torch.argsort(key); del key
src_local = src_pos[order].to(torch.int16).contiguous()                                            # bits of a uint16
want = D.reshape(-1)[(u_of * S + pos)[order]]
del order, u_of, pos, src_pos
w_ent = torch.zeros(E, device=dev)
lds = int(tile_cnt.max()) * 4
ptr = lambda t: ctypes.c_void_p(t.data_ptr())
def run(form, threads):
    rc = P.proto_transpose(ptr(D), S, ptr(lo_sb), ptr(src_local), ptr(tile_ptr), ptr(w_ent), ctypes.c_int64(m), upb, n_jt, C, lds, form,
                           threads, None)
    assert rc == 0, rc
for form, threads in ((0, 1024), (1, 1024), (0, 512), (1, 512), (1, 256)):
  w_ent.zero_(); run(form, threads); torch.cuda.synchronize()
  assert torch.equal(w_ent, want), 'permutation wrong'
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(5): run(form, threads)
  b.record(); torch.cuda.synchronize()
  ms = a.elapsed_time(b) / 5
  print(f'form {form}, {threads} threads: transposition of {E} weights in {C} x {n_jt} tiles (<= {int(tile_cnt.max())} entries, {lds} bytes of LDS): {ms:.2f} ms; '
        f'payload 3 x {E * 4 / 1e9:.1f} GB (read D, index, write) -> {3 * E * 4 / ms / 1e6:.0f} GB/s of payload', flush=True)
