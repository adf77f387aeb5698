#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cmath>
volatile float sink;
int main(){ for(int n: {1,2,4,8}){ auto t0=std::chrono::steady_clock::now(); std::vector<std::thread> p; for(int t=0;t<n;t++) p.emplace_back([]{ float a=0; for(int i=0;i<60000000;i++) a+=sinf(i*1e-3f); sink=a;}); for(auto&t:p) t.join(); printf("%d threads: %.0f ms\n",n,std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now()-t0).count()); } }
