#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <time.h>
#include <sys/mman.h>
int main(){ size_t n=11190402; size_t bytes=n*32; uint32_t*a=mmap(0,bytes,PROT_READ|PROT_WRITE,MAP_PRIVATE|MAP_ANONYMOUS,-1,0); madvise(a,bytes,MADV_HUGEPAGE);
 uint64_t x=88172645463325252ull; for(size_t i=0;i<n;i++){x^=x<<13;x^=x>>7;x^=x<<17;a[i*8]=(uint32_t)(x%n);}
 struct timespec t0,t1; uint32_t idx=0; size_t steps=20000000; clock_gettime(CLOCK_MONOTONIC,&t0);
 for(size_t s=0;s<steps;s++){ idx=a[(size_t)idx*8]; }
 clock_gettime(CLOCK_MONOTONIC,&t1); double dt=(t1.tv_sec-t0.tv_sec)+(t1.tv_nsec-t0.tv_nsec)*1e-9; printf("dependent random load: %.1f ns (idx %u)\n",dt/steps*1e9,idx); return 0;}
