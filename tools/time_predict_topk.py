import torch, time, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from teamoflow_amd import _ops
dev='cuda'
m,n,r=262144,100000,128
U=torch.randn(m,r,device=dev)*0.1; V=torch.randn(n,r,device=dev)*0.1
for _ in range(2):
    _ops.predict_topk(U,V,10,clamp_negatives=True)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(3): _ops.predict_topk(U,V,10,clamp_negatives=True)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/3
print('rows/s', m/dt, 'TF', 2*m*n*r/dt/1e12)
