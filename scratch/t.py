import torch, ctypes, os
torch.cuda.init()
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), 'libt.so'))
x = torch.ones(1000, device='cuda'); y = torch.zeros(1000, device='cuda')
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    rc = lib.tmf_test_axpy(ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_float(2.5), 1000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print('rc', rc, y[:3], y.sum().item())
import subprocess
print(open('/proc/self/maps').read().count('libamdhip64'))
print(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l))
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0))
