// does getrandom() (what os.urandom calls) scale with threads on this host?  g++ -O2 -pthread; prints GB/s for 1..32 threads
// (the randomizer polynomial of code/fast_stark.py:116-117 draws 17 bytes per coefficient: 36 MB at a 2^24 FRI domain)
#include <sys/random.h>
#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>
int main(int argc,char**argv){
  size_t n=17u<<21; std::vector<unsigned char> buf(n);
  for(int k: {1,2,4,8,16,32}){
    auto t0=std::chrono::steady_clock::now();
    std::vector<std::thread> th; size_t per=(n+k-1)/k;
    for(int i=0;i<k;i++) th.emplace_back([&,i]{ size_t a=i*per,b=std::min(n,a+per); while(a<b){ ssize_t r=getrandom(buf.data()+a, std::min<size_t>(b-a,1<<20),0); if(r<=0) abort(); a+=r; } });
    for(auto&t:th) t.join();
    double ms=std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count();
    printf("threads %d: %.2f ms (%.2f GB/s)\n",k,ms,n/ms/1e6);
  }
}
