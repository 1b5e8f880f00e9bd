#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* a, double* r0, double* r1, double* r2, int n){
  int i = blockIdx.x*blockDim.x+threadIdx.x; if(i>=n) return;
  double x=a[i]; double r=__builtin_amdgcn_rcp(x); r0[i]=r;
  r=fma(fma(-x,r,1.0),r,r); r1[i]=r; r=fma(fma(-x,r,1.0),r,r); r2[i]=r;
}
int main(){ int n=1<<20; std::vector<double> h(n); for(int i=0;i<n;i++) h[i]=M_PI*(0.000001+ (double)rand()/RAND_MAX*1200.0);
 double *a,*r0,*r1,*r2; hipMalloc(&a,n*8);hipMalloc(&r0,n*8);hipMalloc(&r1,n*8);hipMalloc(&r2,n*8);
 hipMemcpy(a,h.data(),n*8,hipMemcpyHostToDevice); k<<<n/256,256>>>(a,r0,r1,r2,n);
 std::vector<double> o0(n),o1(n),o2(n); hipMemcpy(o0.data(),r0,n*8,hipMemcpyDeviceToHost);hipMemcpy(o1.data(),r1,n*8,hipMemcpyDeviceToHost);hipMemcpy(o2.data(),r2,n*8,hipMemcpyDeviceToHost);
 double e0=0,e1=0,e2=0; for(int i=0;i<n;i++){ double t=1.0/h[i]; e0=fmax(e0,fabs(o0[i]-t)/t); e1=fmax(e1,fabs(o1[i]-t)/t); e2=fmax(e2,fabs(o2[i]-t)/t);} 
 printf("rcp max rel err: raw %.3e  1NR %.3e  2NR %.3e\n",e0,e1,e2); return 0; }
