#include <hip/hip_runtime.h>
extern "C" __global__ void k_axpy(float* y, const float* x, float a, int n){
  int i = blockIdx.x*blockDim.x+threadIdx.x; if(i<n) y[i] += a*x[i];
}
extern "C" int tmf_test_axpy(float* y, const float* x, float a, int n, void* stream){
  hipLaunchKernelGGL(k_axpy, dim3((n+255)/256), dim3(256), 0, (hipStream_t)stream, y, x, a, n);
  return (int)hipGetLastError();
}
