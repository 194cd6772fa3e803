// Sustained MFMA rate AT THE BOARD'S POWER CAP (measurement only; not part of the library): the training step holds the MI355X at its 1400 W cap
// (tools/power_probe.py) and the clock falls to ~2.0 GHz, so what a K loop delivers is flops per JOULE, not flops per issue slot.  This runs dense
// MFMA loops for seconds each and prints TFLOP/s, board power and the clock held:
//   a  v_mfma_f32_16x16x32_bf16, 4 waves per CU (128 x 128 wave blocks: 64 accumulator tiles), operands in registers
//   b  v_mfma_f32_32x32x16_bf16, same flops per wave, operands in registers (half the instructions, half the VGPR operand reads per flop)
//   c  = a + 16 fragment reads (ds_read_b128) per 64 MFMAs from a random LDS image (the four-wave kernels' ratio)
//   d  = a with 8 waves per CU (128 x 64 blocks) + 24 fragment reads per 64 MFMAs (the eight-wave kernels' ratio)
//   e  = b + the same 16 reads per 32 MFMAs of 32x32x16
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/mfma_power.hip && /tmp/mfma_power [seconds]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>
#include <unistd.h>
#include <limits.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bfrag;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))
void k(const uint32_t* __restrict__ rnd, int steps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 65536 / 4; i += NW * 64) reinterpret_cast<uint32_t*>(smem)[i] = rnd[i];
  __syncthreads();
  constexpr bool M32 = MODE == 1 || MODE == 4;
  constexpr bool LDS = MODE >= 2;
  constexpr int NJ = NW == 8 ? 4 : 8;
  bfrag fa[4], fb[8];
  const char* base = smem + (lane & 15) * 128 + ((lane >> 4) << 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bfrag*>(base + i * 2048);
#pragma unroll
  for (int j = 0; j < 8; ++j) fb[j] = *reinterpret_cast<const bfrag*>(base + 8192 + j * 2048);
  if constexpr (!M32) {
    f32x4 acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if constexpr (LDS) {   // the step's fragment reads, spread over its four sub-phases: 4 A fragments each, NJ B fragments per K half
            const char* st = smem + ((s & 1) << 15) + kk * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bfrag*>(st + (lane & 15) * 128 + ((lane >> 4) << 4) + (half * 4 + i) * 2048);
            if (half == 0) {
#pragma unroll
              for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const bfrag*>(st + 16384 + (lane & 15) * 128 + ((lane >> 4) << 4) + j * 2048);
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[half * 4 + i][j], 0, 0, 0);
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) t += acc[i][j][0] + acc[i][j][3];
    if (t == 1.2345e30f) sink[0] = t;
  } else {
    // 128 x 128 wave block = 4 x 4 tiles of 32 x 32; a 64-deep step = 4 k-chunks of 16 x 16 tiles = 64 MFMAs (same flops as 128 of 16x16x32)
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        if constexpr (LDS) {   // per 16-deep chunk: 4 A + 4 B fragments of 32 rows x 16 k (8 bf16 per lane): 32 reads per step, as in mode c
          const char* st = smem + ((s & 1) << 15) + kc * 32;
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bfrag*>(st + (lane & 31) * 128 + ((lane >> 5) << 4) + i * 4096);
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bfrag*>(st + 16384 + (lane & 31) * 128 + ((lane >> 5) << 4) + j * 4096);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][15];
    if (t == 1.2345e30f) sink[0] = t;
  }
}

static std::string find_card() {
  char bus[64];
  CK(hipDeviceGetPCIBusId(bus, sizeof(bus), 0));
  for (char* p = bus; *p; ++p) *p = (char)tolower(*p);
  std::string want(bus);
  want = want.substr(0, want.rfind('.'));
  DIR* d = opendir("/sys/class/drm");
  std::string found;
  while (dirent* e = d ? readdir(d) : nullptr) {
    if (strncmp(e->d_name, "card", 4) != 0 || strchr(e->d_name, '-')) continue;
    char real[PATH_MAX];
    std::string dev = std::string("/sys/class/drm/") + e->d_name + "/device";
    if (realpath(dev.c_str(), real) && strstr(real, want.c_str())) found = dev;
  }
  if (d) closedir(d);
  return found;
}
static std::string hwmon_file(const std::string& dev, const char* name) {
  std::string h = dev + "/hwmon";
  DIR* d = opendir(h.c_str());
  std::string out;
  while (dirent* e = d ? readdir(d) : nullptr)
    if (strncmp(e->d_name, "hwmon", 5) == 0) out = h + "/" + e->d_name + "/" + name;
  if (d) closedir(d);
  return out;
}
static double read_num(const std::string& f) {
  FILE* fp = fopen(f.c_str(), "r");
  if (!fp) return 0;
  double v = 0;
  if (fscanf(fp, "%lf", &v) != 1) v = 0;
  fclose(fp);
  return v;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 2.0;
  std::vector<uint32_t> h(16384);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((uint32_t)(0x3f80 + (rand() & 0x7f)) << 16) | (uint32_t)(0x3f80 + (rand() & 0x7f)) | ((rand() & 1) << 15) | ((uint32_t)(rand() & 1) << 31);
  uint32_t* rnd; float* sink;
  CK(hipMalloc(&rnd, h.size() * 4)); CK(hipMalloc(&sink, 64));
  CK(hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const std::string dev = find_card();
  const std::string pw = hwmon_file(dev, "power1_input"), fq = hwmon_file(dev, "freq1_input");
  printf("card %s\n", dev.c_str());
  struct Cfg { const char* name; void (*fn)(const uint32_t*, int, float*); int nw; };
  const Cfg cfgs[] = {
      {"a 16x16x32, 4 waves, registers only      ", k<0, 4>, 4}, {"b 32x32x16, 4 waves, registers only      ", k<1, 4>, 4},
      {"c 16x16x32, 4 waves, 32 LDS reads / step ", k<2, 4>, 4}, {"d 16x16x32, 8 waves, 24 LDS reads / step ", k<2, 8>, 8},
      {"e 32x32x16, 4 waves, 32 LDS reads / step ", k<4, 4>, 4}};
  for (const Cfg& c : cfgs) {
    CK(hipFuncSetAttribute((const void*)c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int steps = 20000;
    const double flops = 256.0 * c.nw * (c.nw == 8 ? 64 : 128) * 16384.0 * steps;   // per launch
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(c.fn, dim3(256), dim3(c.nw * 64), 65536, 0, rnd, 2000, sink);
    CK(hipDeviceSynchronize());
    usleep(300000);
    std::atomic<bool> stop{false};
    std::vector<std::pair<double, double>> smp;
    std::thread th([&] { while (!stop) { smp.emplace_back(read_num(pw) / 1e6, read_num(fq) / 1e6); usleep(5000); } });
    int n = 0;
    auto t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, 0));
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
      for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(c.fn, dim3(256), dim3(c.nw * 64), 65536, 0, rnd, steps, sink);
      n += 4;
      CK(hipStreamSynchronize(0));
    }
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    stop = true; th.join();
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    double w = 0, f = 0; int cnt = 0;
    for (size_t i = smp.size() / 2; i < smp.size(); ++i) { w += smp[i].first; f += smp[i].second; ++cnt; }
    w /= cnt ? cnt : 1; f /= cnt ? cnt : 1;
    const double tf = flops * n / (ms * 1e-3) / 1e12;
    printf("%s %7.1f TFLOP/s at %5.0f W, %5.0f MHz  -> %5.2f pJ per flop, %4.1f %% of the peak at that clock\n", c.name, tf, w, f, w / (tf * 1e12) * 1e12,
           100.0 * tf / (256 * 4 * 1024.0 * f * 1e6 / 1e12));
  }
  return 0;
}
