// plan_host — a host for the AVSyncD denoising path that contains no Python and no torch: it loads a launch-plan bundle
// (include/avsd.h "launch plans", written by asva_amd/plan.py), allocates and binds the buffers with the HIP runtime, and
// runs the per-clip conditioning, the denoising loop (UNet forward plan + avsd_guided_step with the exported scheduler
// table) and the VAE decode through the C ABI of libavsd_hip.so alone.
//
//   plan_host <libavsd_hip.so> <bundle.plan> <program.txt>
// program.txt, one command per line:
//   load <region> <file>                       upload a file into an INPUT region
//   run <plan>                                 avsd_plan_run
//   denoise <steps.bin> <latents> <x> <t> <noise> <n_branch> <g> <g2> <B> <C> <F> <HW>
//                                              the loop of pipeline_audio_cond_animation.py:325-365 on buffer <latents>
//                                              (f32, B x C x F x HW): per step  x <- latents, t <- steps[i].t, run "forward",
//                                              guidance + scheduler update in place (avsd_guided_step), frame 0 pinned
//                                              PLAN_HOST_GRAPH=1: the forward plan is captured once into a hipGraph and
//                                              replayed (avsd_plan_run only issues launches on the stream it is given)
//   copy <src> <dst> <bytes>                   device-to-device between regions
//   save <region> <file>                       download a region to a file
// <latents>, <x>, <t>, <noise> and the names after load / save / copy are REGION names of the bundle.
// Build: hipcc -O2 -std=c++17 tools/plan_host.cpp -Iinclude -ldl -o asva_amd/plan_host   (asva_amd/build.py does it)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "avsd.h"

#define HIP_OK(x)                                                                      \
  do {                                                                                 \
    hipError_t e__ = (x);                                                              \
    if (e__ != hipSuccess) {                                                           \
      fprintf(stderr, "plan_host: %s: %s\n", #x, hipGetErrorString(e__));             \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)

struct Api {
  decltype(&avsd_last_error) last_error;
  decltype(&avsd_device_info) device_info;
  decltype(&avsd_plan_bundle_load) load;
  decltype(&avsd_plan_bundle_free) free_;
  decltype(&avsd_plan_bundle_num_buffers) num_buffers;
  decltype(&avsd_plan_bundle_buffer_bytes) buffer_bytes;
  decltype(&avsd_plan_bundle_bind) bind;
  decltype(&avsd_plan_bundle_num_regions) num_regions;
  decltype(&avsd_plan_bundle_region) region;
  decltype(&avsd_plan_bundle_find_region) find_region;
  decltype(&avsd_plan_bundle_find_plan) find_plan;
  decltype(&avsd_plan_num_calls) num_calls;
  decltype(&avsd_plan_run) run;
  decltype(&avsd_guided_step) guided_step;
  decltype(&avsd_copy) copy;
};

template <class T>
static void sym(void* h, const char* name, T& fn) {
  fn = reinterpret_cast<T>(dlsym(h, name));
  if (!fn) {
    fprintf(stderr, "plan_host: %s not exported by the library\n", name);
    exit(2);
  }
}

// one entry of the scheduler table (asva_amd/plan.py: export_steps; schedulers.StepPlan)
struct Step {
  float t, ca, cb, w_cur;
  int32_t store_slot, n_hist, save_sample, use_saved;
  int32_t hist_idx[4];
  float hist_w[4];
};

static std::vector<unsigned char> read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    fprintf(stderr, "plan_host: cannot read %s\n", path.c_str());
    exit(2);
  }
  return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc != 4) {
    fprintf(stderr, "usage: plan_host <libavsd_hip.so> <bundle.plan> <program.txt>\n");
    return 2;
  }
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    fprintf(stderr, "plan_host: %s\n", dlerror());
    return 2;
  }
  Api api;
  sym(h, "avsd_last_error", api.last_error);
  sym(h, "avsd_device_info", api.device_info);
  sym(h, "avsd_plan_bundle_load", api.load);
  sym(h, "avsd_plan_bundle_free", api.free_);
  sym(h, "avsd_plan_bundle_num_buffers", api.num_buffers);
  sym(h, "avsd_plan_bundle_buffer_bytes", api.buffer_bytes);
  sym(h, "avsd_plan_bundle_bind", api.bind);
  sym(h, "avsd_plan_bundle_num_regions", api.num_regions);
  sym(h, "avsd_plan_bundle_region", api.region);
  sym(h, "avsd_plan_bundle_find_region", api.find_region);
  sym(h, "avsd_plan_bundle_find_plan", api.find_plan);
  sym(h, "avsd_plan_num_calls", api.num_calls);
  sym(h, "avsd_plan_run", api.run);
  sym(h, "avsd_guided_step", api.guided_step);
  sym(h, "avsd_copy", api.copy);
#define AVSD_OK_OR_DIE(x)                                                              \
  do {                                                                                 \
    if ((x) != AVSD_OK) {                                                              \
      fprintf(stderr, "plan_host: %s: %s\n", #x, api.last_error());                   \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

  char arch[64];
  int ncu = 0;
  AVSD_OK_OR_DIE(api.device_info(arch, sizeof(arch), &ncu));
  const std::string bundle_path = argv[2];
  avsd_plan_bundle* b = nullptr;
  AVSD_OK_OR_DIE(api.load(bundle_path.c_str(), &b));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));

  // allocate and bind every buffer (the recording run's allocator segments), zero-filled; then upload the CONST regions
  const int nb = api.num_buffers(b);
  std::vector<unsigned char*> dev(nb, nullptr);
  int64_t total = 0;
  for (int i = 0; i < nb; ++i) {
    const int64_t bytes = api.buffer_bytes(b, i);
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&dev[i]), (size_t)bytes));
    HIP_OK(hipMemset(dev[i], 0, (size_t)bytes));
    AVSD_OK_OR_DIE(api.bind(b, i, dev[i]));
    total += bytes;
  }
  struct Reg {
    unsigned char* ptr;
    int64_t bytes;
  };
  auto reg = [&](const std::string& name) {
    const int j = api.find_region(b, name.c_str());
    if (j < 0) {
      fprintf(stderr, "plan_host: no region named %s\n", name.c_str());
      exit(1);
    }
    int bi, kind;
    int64_t off, bytes;
    api.region(b, j, nullptr, &bi, &off, &bytes, &kind);
    return Reg{dev[bi] + off, bytes};
  };
  for (int j = 0; j < api.num_regions(b); ++j) {
    const char* name;
    int bi, kind;
    int64_t off, bytes;
    AVSD_OK_OR_DIE(api.region(b, j, &name, &bi, &off, &bytes, &kind));
    if (kind != AVSD_REGION_CONST) continue;
    const std::vector<unsigned char> data = read_file(bundle_path + ".d/" + name + ".bin");
    if ((int64_t)data.size() != bytes) {
      fprintf(stderr, "plan_host: %s.bin has %zu bytes, the region has %lld\n", name, data.size(), (long long)bytes);
      return 1;
    }
    HIP_OK(hipMemcpy(dev[bi] + off, data.data(), data.size(), hipMemcpyHostToDevice));
  }
  printf("plan_host: %s, %d CUs; %d buffers, %.1f MB bound, %d regions\n", arch, ncu, nb, total / 1e6, api.num_regions(b));

  auto plan = [&](const std::string& name) {
    const int k = api.find_plan(b, name.c_str());
    if (k < 0) {
      fprintf(stderr, "plan_host: no plan named %s\n", name.c_str());
      exit(1);
    }
    return k;
  };

  std::ifstream prog(argv[3]);
  std::string line;
  while (std::getline(prog, line)) {
    std::istringstream ss(line);
    std::string cmd;
    if (!(ss >> cmd) || cmd[0] == '#') continue;
    if (cmd == "load") {
      std::string name, file;
      ss >> name >> file;
      const Reg r = reg(name);
      const std::vector<unsigned char> data = read_file(file);
      if ((int64_t)data.size() != r.bytes) {
        fprintf(stderr, "plan_host: %s has %zu bytes, region %s has %lld\n", file.c_str(), data.size(), name.c_str(), (long long)r.bytes);
        return 1;
      }
      HIP_OK(hipMemcpyAsync(r.ptr, data.data(), data.size(), hipMemcpyHostToDevice, stream));
      HIP_OK(hipStreamSynchronize(stream));
    } else if (cmd == "run") {
      std::string name;
      ss >> name;
      const int k = plan(name);
      AVSD_OK_OR_DIE(api.run(b, k, stream));
      printf("plan_host: ran %s (%d launches)\n", name.c_str(), api.num_calls(b, k));
    } else if (cmd == "copy") {
      std::string s, d;
      long long bytes;
      ss >> s >> d >> bytes;
      AVSD_OK_OR_DIE(api.copy(reg(s).ptr, reg(d).ptr, bytes, 1, stream));
    } else if (cmd == "denoise") {
      std::string file, lat, x, t, noise;
      int n_branch, B, Cc, Fr, HW;
      float g, g2;
      ss >> file >> lat >> x >> t >> noise >> n_branch >> g >> g2 >> B >> Cc >> Fr >> HW;
      const std::vector<unsigned char> raw = read_file(file);
      const size_t n = raw.size() / sizeof(Step);
      const Step* steps = reinterpret_cast<const Step*>(raw.data());
      const int64_t lat_bytes = (int64_t)B * Cc * Fr * HW * 4;
      float *hist = nullptr, *saved = nullptr;
      HIP_OK(hipMalloc(&hist, (size_t)(4 * lat_bytes)));
      HIP_OK(hipMemset(hist, 0, (size_t)(4 * lat_bytes)));
      HIP_OK(hipMalloc(&saved, (size_t)lat_bytes));
      float* latents = reinterpret_cast<float*>(reg(lat).ptr);
      unsigned char *xp = reg(x).ptr, *tp = reg(t).ptr;
      const float* np_ = reinterpret_cast<const float*>(reg(noise).ptr);
      if (reg(lat).bytes != lat_bytes || reg(x).bytes != lat_bytes) {
        fprintf(stderr, "plan_host: denoise: B x C x F x HW does not match the latent regions\n");
        return 1;
      }
      const int kf = plan("forward");
      const char* genv = getenv("PLAN_HOST_GRAPH");
      hipGraphExec_t gexec = nullptr;
      if (genv && genv[0] == '1') {
        AVSD_OK_OR_DIE(api.run(b, kf, stream));   // once eagerly: lazy module loading does not belong in a capture
        HIP_OK(hipStreamSynchronize(stream));
        hipGraph_t graph;
        HIP_OK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        AVSD_OK_OR_DIE(api.run(b, kf, stream));
        HIP_OK(hipStreamEndCapture(stream, &graph));
        HIP_OK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
        HIP_OK(hipGraphDestroy(graph));
      }
      hipEvent_t e0, e1;
      HIP_OK(hipEventCreate(&e0));
      HIP_OK(hipEventCreate(&e1));
      HIP_OK(hipEventRecord(e0, stream));
      for (size_t i = 0; i < n; ++i) {
        const Step& s = steps[i];
        AVSD_OK_OR_DIE(api.copy(latents, xp, lat_bytes, 1, stream));
        HIP_OK(hipMemcpyAsync(tp, &s.t, 4, hipMemcpyHostToDevice, stream));
        if (gexec) HIP_OK(hipGraphLaunch(gexec, stream));
        else AVSD_OK_OR_DIE(api.run(b, kf, stream));
        if (s.save_sample) AVSD_OK_OR_DIE(api.copy(latents, saved, lat_bytes, 1, stream));
        AVSD_OK_OR_DIE(api.guided_step(np_, n_branch, g, g2, hist, s.store_slot, s.w_cur,
                                       s.n_hist ? s.hist_idx : nullptr, s.n_hist ? s.hist_w : nullptr, s.n_hist,
                                       s.use_saved ? saved : latents, latents, s.ca, s.cb, B, Cc, Fr, HW, stream));
      }
      HIP_OK(hipEventRecord(e1, stream));
      HIP_OK(hipStreamSynchronize(stream));
      float ms = 0.f;
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      if (gexec) HIP_OK(hipGraphExecDestroy(gexec));
      HIP_OK(hipFree(hist));
      HIP_OK(hipFree(saved));
      printf("plan_host: %zu denoising steps, %.3f ms per step (%s)\n", n, ms / (n ? n : 1), gexec ? "hipGraph replay" : "eager launches");
    } else if (cmd == "save") {
      std::string name, file;
      ss >> name >> file;
      const Reg r = reg(name);
      HIP_OK(hipStreamSynchronize(stream));
      std::vector<unsigned char> data((size_t)r.bytes);
      HIP_OK(hipMemcpy(data.data(), r.ptr, data.size(), hipMemcpyDeviceToHost));
      std::ofstream f(file, std::ios::binary);
      f.write(reinterpret_cast<const char*>(data.data()), (std::streamsize)data.size());
    } else {
      fprintf(stderr, "plan_host: unknown command %s\n", cmd.c_str());
      return 1;
    }
  }
  HIP_OK(hipStreamSynchronize(stream));
  for (unsigned char* p : dev) HIP_OK(hipFree(p));
  api.free_(b);
  printf("plan_host: done\n");
  return 0;
}
