// Host -> device bandwidth of the trace-row upload (60 MiB per 2^20-row Add proof) by kind of host memory and NUMA node:
// hipHostMalloc (default / non-coherent / NumaUser bound to each node), malloc + hipHostRegister on each node, pageable on
// each node; copies by hipMemcpyAsync (1, 2, 4 streams) and by a kernel that reads the host buffer directly.
// Prints where the GPU, the calling thread and each buffer's pages live.  Build: hipcc --offload-arch=gfx950 -O2
// tools/microbench_h2d.hip -o tools/bin/mb_h2d
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static long sys_move_pages(int pid, unsigned long n, void** pages, const int* nodes, int* status, int flags) {
  return syscall(SYS_move_pages, pid, n, pages, nodes, status, flags);
}
static long sys_mbind(void* addr, unsigned long len, int mode, const unsigned long* mask, unsigned long maxnode, unsigned flags) {
  return syscall(SYS_mbind, addr, len, mode, mask, maxnode, flags);
}
static long sys_set_mempolicy(int mode, const unsigned long* mask, unsigned long maxnode) {
  return syscall(SYS_set_mempolicy, mode, mask, maxnode);
}
static int page_node(void* p) {
  void* pg = (void*)((uintptr_t)p & ~(uintptr_t)4095);
  int st = -99;
  if (sys_move_pages(0, 1, &pg, nullptr, &st, 0) != 0) return -98;
  return st;
}
static std::string node_hist(void* p, size_t bytes) {
  int cnt[16] = {0};
  int other = 0;
  for (size_t off = 0; off < bytes; off += bytes / 16) {
    int n = page_node((char*)p + off);
    if (n >= 0 && n < 16) cnt[n]++; else other++;
  }
  std::string s;
  for (int i = 0; i < 16; ++i) if (cnt[i]) s += "node" + std::to_string(i) + ":" + std::to_string(cnt[i]) + " ";
  if (other) s += "unknown:" + std::to_string(other);
  return s;
}
static int n_nodes() {
  int n = 0;
  for (int i = 0; i < 16; ++i) {
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d", i);
    if (access(path, F_OK) == 0) n = i + 1;
  }
  return n ? n : 1;
}

__global__ void k_read_host(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) dst[i] = src[i];
}

static double time_copy(void* dev[], const void* host, size_t bytes, int streams, hipStream_t* st, int reps) {
  // `streams` concurrent copies of the same host buffer into different device buffers
  for (int s = 0; s < streams; ++s) CK(hipMemcpyAsync(dev[s], host, bytes, hipMemcpyHostToDevice, st[s]));
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r)
    for (int s = 0; s < streams; ++s) CK(hipMemcpyAsync(dev[s], host, bytes, hipMemcpyHostToDevice, st[s]));
  CK(hipDeviceSynchronize());
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return (double)bytes * reps * streams / sec / 1e9;
}
static double time_kernel(void* dev, const void* host, size_t bytes, int blocks, hipStream_t st, int reps) {
  hipLaunchKernelGGL(k_read_host, dim3(blocks), dim3(256), 0, st, (const uint4*)host, (uint4*)dev, bytes / 16);
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(k_read_host, dim3(blocks), dim3(256), 0, st, (const uint4*)host, (uint4*)dev, bytes / 16);
  CK(hipDeviceSynchronize());
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return (double)bytes * reps / sec / 1e9;
}

int main() {
  const size_t bytes = 60u << 20;
  CK(hipSetDevice(0));
  char bus[64] = {0};
  CK(hipDeviceGetPCIBusId(bus, sizeof bus, 0));
  for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
  char path[160];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
  int gpu_node = -1;
  if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &gpu_node) != 1) gpu_node = -1; fclose(f); }
  const int nodes = n_nodes();
  int cpu = sched_getcpu();
  cpu_set_t set;
  CPU_ZERO(&set);
  sched_getaffinity(0, sizeof set, &set);
  printf("GPU %s numa_node %d; host numa nodes %d; this thread on cpu %d, affinity %d cpus\n", bus, gpu_node, nodes, cpu, CPU_COUNT(&set));
  for (int n = 0; n < nodes; ++n) {
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", n);
    char buf[256] = {0};
    if (FILE* f = fopen(path, "r")) { if (!fgets(buf, sizeof buf, f)) buf[0] = 0; fclose(f); }
    printf("  node%d cpus %s", n, buf[0] ? buf : "?\n");
  }
  hipStream_t st[4];
  void* dev[4];
  for (int s = 0; s < 4; ++s) { CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking)); CK(hipMalloc(&dev[s], bytes)); }

  auto report = [&](const char* name, void* host, bool kernel_ok) {
    printf("%-44s pages %-22s", name, node_hist(host, bytes).c_str());
    for (int ns : {1, 2, 4}) printf("  memcpy x%d %5.1f GB/s", ns, time_copy(dev, host, bytes, ns, st, 12));
    if (kernel_ok) for (int bl : {64, 256, 1024}) printf("  kernel %4d blocks %5.1f GB/s", bl, time_kernel(dev[0], host, bytes, bl, st[0], 12));
    printf("\n");
    fflush(stdout);
  };
  {
    void* p = nullptr;
    CK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    memset(p, 1, bytes);
    report("hipHostMalloc default", p, true);
    CK(hipHostFree(p));
    CK(hipHostMalloc(&p, bytes, hipHostMallocNonCoherent));
    memset(p, 1, bytes);
    report("hipHostMalloc non-coherent", p, true);
    CK(hipHostFree(p));
  }
  for (int n = 0; n < nodes; ++n) {
    unsigned long mask = 1ul << n;
    if (sys_set_mempolicy(2 /* MPOL_BIND */, &mask, 64) != 0) { printf("set_mempolicy(node%d) failed\n", n); continue; }
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocNumaUser);
    sys_set_mempolicy(0, nullptr, 0);
    if (e != hipSuccess) { printf("hipHostMalloc NumaUser node%d: %s\n", n, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    memset(p, 1, bytes);
    std::string nm = "hipHostMalloc NumaUser bound to node" + std::to_string(n);
    report(nm.c_str(), p, true);
    CK(hipHostFree(p));
  }
  for (int n = 0; n < nodes; ++n) {
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    unsigned long mask = 1ul << n;
    if (sys_mbind(p, bytes, 2, &mask, 64, 0) != 0) { printf("mbind(node%d) failed\n", n); munmap(p, bytes); continue; }
    memset(p, 1, bytes);
    std::string nm = "pageable (mmap) on node" + std::to_string(n);
    report(nm.c_str(), p, false);
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) {
      nm = "mmap + hipHostRegister on node" + std::to_string(n);
      report(nm.c_str(), p, true);
      CK(hipHostUnregister(p));
    } else {
      (void)hipGetLastError();
      printf("hipHostRegister failed on node%d\n", n);
    }
    munmap(p, bytes);
  }
  {
    void* p = malloc(bytes);
    memset(p, 1, bytes);
    report("malloc, first touch by this thread", p, false);
    free(p);
  }
  return 0;
}
