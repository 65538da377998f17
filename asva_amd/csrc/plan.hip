// Launch plans (include/avsd.h, "launch plans"): a bundle file holds a buffer table and named call lists; every device
// pointer of a recorded call is (buffer, byte offset).  avsd_plan_run re-issues the calls through the very entry points
// the Python host used, so a host without Python / torch runs the UNet forward, the conditioning projections or the VAE
// decode with bit-identical results.  Host code only: nothing here touches the device except through those entry points.
//
// File layout (little-endian), written by asva_amd/plan.py:
//   "AVSDPLN1"  u32 abi  char precision[8]
//   u32 n_buffers { i64 bytes }                                              (the recording run's allocator segments)
//   u32 n_regions { u32 len, name, i32 buffer, i64 offset, i64 bytes, u32 kind }   (named places inside them)
//   u32 n_plans   { u32 len, name, u32 n_calls { u32 len, entry-point name, u32 n_args { u8 tag, payload } } }
//   tags: 'i' i64 | 'f' f64 | 'p' i32 buffer (-1 = NULL) + i64 offset | 'S' the stream of avsd_plan_run |
//         'h' u32 n + n bytes: a host array passed by pointer (n = 0: NULL) |
//         's' u32 n + n bytes + u32 n_reloc { u32 field offset, i32 buffer, i64 offset }: a descriptor passed by pointer
#include "avsd_common.h"

#include <stdio.h>
#include <string.h>

#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

struct RArg {   // one resolved argument
  int64_t i = 0;
  double f = 0.0;
  void* p = nullptr;
};

template <class T>
T conv(const RArg& a) {
  if constexpr (std::is_pointer_v<T>) return reinterpret_cast<T>(a.p);
  else if constexpr (std::is_floating_point_v<T>) return static_cast<T>(a.f);
  else return static_cast<T>(a.i);
}
template <class... A, size_t... I>
int call_impl(int (*fn)(A...), const RArg* a, std::index_sequence<I...>) {
  return fn(conv<A>(a[I])...);
}
template <class... A>
int call(int (*fn)(A...), const RArg* a, int n, const char* name) {
  if (n != (int)sizeof...(A)) {
    avsd_set_error("plan: %s recorded with %d arguments, this library takes %d", name, n, (int)sizeof...(A));
    return AVSD_EINVAL;
  }
  return call_impl(fn, a, std::index_sequence_for<A...>{});
}

typedef int (*thunk_t)(const RArg*, int);
struct Entry {
  const char* name;
  thunk_t thunk;
  size_t desc_bytes;   // size every 's' (descriptor) argument of this entry point must have; 0 = takes none
};
#define AVSD_PLAN_ENTRY(fn) {#fn, [](const RArg* a, int n) { return call(&fn, a, n, #fn); }, 0}
#define AVSD_PLAN_ENTRY_DESC(fn, T) {#fn, [](const RArg* a, int n) { return call(&fn, a, n, #fn); }, sizeof(T)}
// every entry point that launches work (the queries and the plan API itself are not recordable)
const Entry kEntries[] = {
    AVSD_PLAN_ENTRY_DESC(avsd_gemm_bf16, avsd_gemm_desc), AVSD_PLAN_ENTRY_DESC(avsd_cross_attention_block, avsd_xattn_desc), AVSD_PLAN_ENTRY(avsd_linear_small_m),
    AVSD_PLAN_ENTRY(avsd_groupnorm_stats),  AVSD_PLAN_ENTRY(avsd_groupnorm_apply),       AVSD_PLAN_ENTRY(avsd_layernorm),
    AVSD_PLAN_ENTRY(avsd_softmax_rows),     AVSD_PLAN_ENTRY(avsd_attention),             AVSD_PLAN_ENTRY(avsd_attention_fp8),
    AVSD_PLAN_ENTRY(avsd_temporal_attention), AVSD_PLAN_ENTRY(avsd_ncfhw_to_rows),       AVSD_PLAN_ENTRY(avsd_rows_to_ncfhw),
    AVSD_PLAN_ENTRY(avsd_timestep_embedding), AVSD_PLAN_ENTRY(avsd_guided_step),         AVSD_PLAN_ENTRY(avsd_vae_postprocess),
    AVSD_PLAN_ENTRY(avsd_vae_postprocess_u8), AVSD_PLAN_ENTRY(avsd_kaldi_fbank),         AVSD_PLAN_ENTRY(avsd_patchify),
    AVSD_PLAN_ENTRY(avsd_vit_tokens),       AVSD_PLAN_ENTRY(avsd_copy),                  AVSD_PLAN_ENTRY(avsd_xattn_pack_kv),
    // split-precision ("x2") entry points
    AVSD_PLAN_ENTRY(avsd_linear_small_m_x2), AVSD_PLAN_ENTRY(avsd_groupnorm_stats_x2),   AVSD_PLAN_ENTRY(avsd_groupnorm_apply_x2),
    AVSD_PLAN_ENTRY(avsd_layernorm_x2),     AVSD_PLAN_ENTRY(avsd_attention_x2),          AVSD_PLAN_ENTRY(avsd_temporal_attention_x2),
    AVSD_PLAN_ENTRY(avsd_ncfhw_to_rows_x2), AVSD_PLAN_ENTRY(avsd_split_f32),             AVSD_PLAN_ENTRY(avsd_vae_postprocess_x2),
    AVSD_PLAN_ENTRY(avsd_vae_postprocess_u8_x2), AVSD_PLAN_ENTRY(avsd_softmax_rows_x2),
    AVSD_PLAN_ENTRY(avsd_groupnorm_fused),  AVSD_PLAN_ENTRY(avsd_groupnorm_fused_x2),
    AVSD_PLAN_ENTRY(avsd_ln_fold),
};

struct Reloc {
  uint32_t field_off;
  int32_t buf;
  int64_t off;
};
struct Arg {
  uint8_t tag = 0;
  int64_t i = 0;
  double f = 0.0;
  int32_t buf = -1;
  int64_t off = 0;
  std::vector<unsigned char> blob;   // 'h' host array / 's' descriptor image (patched in place at resolve time)
  std::vector<Reloc> relocs;
};
struct Call {
  thunk_t thunk = nullptr;
  std::string fn;
  std::vector<Arg> args;
  std::vector<RArg> resolved;
};
struct Buffer {
  int64_t bytes = 0;
  void* ptr = nullptr;
};
struct Region {
  std::string name;
  int32_t buf = -1;
  int64_t off = 0, bytes = 0;
  int kind = 0;
};
struct Plan {
  std::string name;
  std::vector<Call> calls;
  bool resolved = false;
};

struct Reader {
  FILE* f;
  bool ok = true;
  void raw(void* dst, size_t n) {
    if (ok && n && fread(dst, 1, n, f) != n) ok = false;
  }
  template <class T>
  T get() {
    T v{};
    raw(&v, sizeof(T));
    return v;
  }
  std::string str() {
    const uint32_t n = get<uint32_t>();
    if (!ok || n > (1u << 20)) { ok = false; return std::string(); }
    std::string s(n, '\0');
    raw(&s[0], n);
    return s;
  }
};

}  // namespace

struct avsd_plan_bundle {
  std::vector<Buffer> buffers;
  std::vector<Region> regions;
  std::vector<Plan> plans;
};

namespace {

// pointer of (buffer, offset), or an error naming what is missing
int resolve_ptr(const avsd_plan_bundle& b, int32_t buf, int64_t off, const char* fn, void** out) {
  if (buf < 0) { *out = nullptr; return AVSD_OK; }
  AVSD_REQUIRE(buf < (int)b.buffers.size(), "plan: %s refers to buffer %d of %d", fn, buf, (int)b.buffers.size());
  const Buffer& bf = b.buffers[buf];
  AVSD_REQUIRE(bf.ptr != nullptr, "plan: buffer %d (%lld bytes) is not bound", buf, (long long)bf.bytes);
  AVSD_REQUIRE(off >= 0 && off < bf.bytes, "plan: %s reads offset %lld of buffer %d (%lld bytes)", fn, (long long)off, buf,
               (long long)bf.bytes);
  *out = static_cast<unsigned char*>(bf.ptr) + off;
  return AVSD_OK;
}

int resolve(avsd_plan_bundle& b, Plan& p) {
  for (Call& c : p.calls) {
    c.resolved.assign(c.args.size(), RArg());
    for (size_t k = 0; k < c.args.size(); ++k) {
      Arg& a = c.args[k];
      RArg& r = c.resolved[k];
      switch (a.tag) {
        case 'i': r.i = a.i; break;
        case 'f': r.f = a.f; break;
        case 'S': break;   // filled per run
        case 'p': {
          const int rc = resolve_ptr(b, a.buf, a.off, c.fn.c_str(), &r.p);
          if (rc != AVSD_OK) return rc;
          break;
        }
        case 'h': r.p = a.blob.empty() ? nullptr : a.blob.data(); break;
        case 's': {
          for (const Reloc& rl : a.relocs) {
            void* ptr = nullptr;
            const int rc = resolve_ptr(b, rl.buf, rl.off, c.fn.c_str(), &ptr);
            if (rc != AVSD_OK) return rc;
            AVSD_REQUIRE(rl.field_off + sizeof(void*) <= a.blob.size(), "plan: %s descriptor relocation out of range", c.fn.c_str());
            memcpy(a.blob.data() + rl.field_off, &ptr, sizeof(void*));
          }
          r.p = a.blob.data();
          break;
        }
        default: AVSD_REQUIRE(false, "plan: unknown argument tag %d in %s", (int)a.tag, c.fn.c_str());
      }
    }
  }
  p.resolved = true;
  return AVSD_OK;
}

}  // namespace

static int plan_bundle_load_impl(const char* path, avsd_plan_bundle** out);

extern "C" int avsd_plan_bundle_load(const char* path, avsd_plan_bundle** out) {
  try {                         // nothing may propagate across the C boundary (std::bad_alloc on a hostile count)
    return plan_bundle_load_impl(path, out);
  } catch (...) {
    if (out) *out = nullptr;
    avsd_set_error("plan_bundle_load: %s: out of memory or corrupt file", path ? path : "(null)");
    return AVSD_EINVAL;
  }
}

static int plan_bundle_load_impl(const char* path, avsd_plan_bundle** out) {
  AVSD_REQUIRE(path && out, "plan_bundle_load: null argument");
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  AVSD_REQUIRE(f != nullptr, "plan_bundle_load: cannot open %s", path);
  Reader rd{f};
  char magic[8];
  rd.raw(magic, 8);
  const uint32_t abi = rd.get<uint32_t>();
  char prec[8];
  rd.raw(prec, 8);
  prec[7] = 0;
  struct Guard {            // closes the file / frees the bundle on every exit path, exceptions included
    FILE* f; avsd_plan_bundle* b;
    ~Guard() { if (f) fclose(f); delete b; }
  } guard{f, new avsd_plan_bundle()};
  avsd_plan_bundle* b = guard.b;
  auto fail = [&](const char* why) {
    avsd_set_error("plan_bundle_load: %s: %s", path, why);
    return AVSD_EINVAL;
  };
  if (!rd.ok || memcmp(magic, "AVSDPLN1", 8) != 0) return fail("not a plan bundle");
  if (abi != AVSD_ABI_VERSION) return fail("recorded against another ABI version of the library");
  if (strcmp(prec, AVSD_PRECISION_NAME) != 0) return fail("recorded for the other storage precision (bf16 / fp16 library)");
  const uint32_t nb = rd.get<uint32_t>();
  if (!rd.ok || nb > (1u << 20)) return fail("bad buffer table");
  b->buffers.resize(nb);
  for (Buffer& bf : b->buffers) bf.bytes = rd.get<int64_t>();
  const uint32_t nr = rd.get<uint32_t>();
  if (!rd.ok || nr > (1u << 20)) return fail("bad region table");
  b->regions.resize(nr);
  for (Region& r : b->regions) {
    r.name = rd.str();
    r.buf = rd.get<int32_t>();
    r.off = rd.get<int64_t>();
    r.bytes = rd.get<int64_t>();
    r.kind = (int)rd.get<uint32_t>();
    if (!rd.ok || r.buf < 0 || r.buf >= (int)nb || r.off < 0 || r.bytes < 0 || r.off > b->buffers[r.buf].bytes ||
        r.bytes > b->buffers[r.buf].bytes - r.off)
      return fail("a region lies outside its buffer");
  }
  const uint32_t np = rd.get<uint32_t>();
  if (!rd.ok || np > (1u << 16)) return fail("bad plan table");
  b->plans.resize(np);
  for (Plan& p : b->plans) {
    p.name = rd.str();
    const uint32_t nc = rd.get<uint32_t>();
    if (!rd.ok || nc > (1u << 20)) return fail("bad call count");
    p.calls.resize(nc);
    for (Call& c : p.calls) {
      c.fn = rd.str();
      size_t desc_bytes = 0;
      for (const Entry& e : kEntries)
        if (c.fn == e.name) { c.thunk = e.thunk; desc_bytes = e.desc_bytes; }
      if (!rd.ok || !c.thunk) return fail("a call names an entry point this library does not record");
      const uint32_t na = rd.get<uint32_t>();
      if (!rd.ok || na > 64) return fail("bad argument count");
      c.args.resize(na);
      for (Arg& a : c.args) {
        a.tag = rd.get<uint8_t>();
        if (!rd.ok) return fail("truncated file");
        if (a.tag == 'i') a.i = rd.get<int64_t>();
        else if (a.tag == 'f') a.f = rd.get<double>();
        else if (a.tag == 'p') { a.buf = rd.get<int32_t>(); a.off = rd.get<int64_t>(); }
        else if (a.tag == 'h' || a.tag == 's') {
          const uint32_t n = rd.get<uint32_t>();
          if (!rd.ok || n > (1u << 20)) return fail("bad blob");
          a.blob.resize(n);
          rd.raw(a.blob.data(), n);
          if (a.tag == 's') {
            if (n != desc_bytes) return fail("a descriptor argument does not have the size its entry point reads");
            const uint32_t nr = rd.get<uint32_t>();
            if (!rd.ok || nr > 256) return fail("bad relocation count");
            a.relocs.resize(nr);
            for (Reloc& rl : a.relocs) {
              rl.field_off = rd.get<uint32_t>(); rl.buf = rd.get<int32_t>(); rl.off = rd.get<int64_t>();
              if (!rd.ok || (size_t)rl.field_off + sizeof(void*) > (size_t)n) return fail("a descriptor relocation lies outside the descriptor");
            }
          } else if (c.fn == "avsd_guided_step" && n != 0 && n != 16) {
            return fail("avsd_guided_step history tables are 4 entries of 4 bytes");   // the entry point reads up to n_hist <= 4
          }
        } else if (a.tag != 'S') return fail("unknown argument tag");
      }
    }
  }
  if (!rd.ok) return fail("truncated file");
  guard.b = nullptr;
  *out = b;
  return AVSD_OK;
}

extern "C" void avsd_plan_bundle_free(avsd_plan_bundle* b) { delete b; }

extern "C" int avsd_plan_bundle_num_buffers(const avsd_plan_bundle* b) { return b ? (int)b->buffers.size() : 0; }

extern "C" int64_t avsd_plan_bundle_buffer_bytes(const avsd_plan_bundle* b, int i) {
  return (b && i >= 0 && i < (int)b->buffers.size()) ? b->buffers[i].bytes : -1;
}

extern "C" int avsd_plan_bundle_bind(avsd_plan_bundle* b, int i, void* device_ptr) {
  AVSD_REQUIRE(b && i >= 0 && i < (int)b->buffers.size(), "plan_bundle_bind: bad index");
  AVSD_REQUIRE(device_ptr != nullptr && (uintptr_t)device_ptr % 256 == 0, "plan_bundle_bind: buffer %d needs a 256-byte-aligned device pointer", i);
  b->buffers[i].ptr = device_ptr;
  for (Plan& p : b->plans) p.resolved = false;
  return AVSD_OK;
}

extern "C" int avsd_plan_bundle_num_regions(const avsd_plan_bundle* b) { return b ? (int)b->regions.size() : 0; }

extern "C" int avsd_plan_bundle_region(const avsd_plan_bundle* b, int j, const char** name, int* buffer, int64_t* offset, int64_t* bytes,
                                       int* kind) {
  AVSD_REQUIRE(b && j >= 0 && j < (int)b->regions.size(), "plan_bundle_region: bad index");
  const Region& r = b->regions[j];
  if (name) *name = r.name.c_str();
  if (buffer) *buffer = r.buf;
  if (offset) *offset = r.off;
  if (bytes) *bytes = r.bytes;
  if (kind) *kind = r.kind;
  return AVSD_OK;
}

extern "C" int avsd_plan_bundle_find_region(const avsd_plan_bundle* b, const char* name) {
  if (!b || !name) return -1;
  for (size_t j = 0; j < b->regions.size(); ++j)
    if (b->regions[j].name == name) return (int)j;
  return -1;
}

extern "C" int avsd_plan_bundle_num_plans(const avsd_plan_bundle* b) { return b ? (int)b->plans.size() : 0; }

extern "C" const char* avsd_plan_bundle_plan_name(const avsd_plan_bundle* b, int k) {
  return (b && k >= 0 && k < (int)b->plans.size()) ? b->plans[k].name.c_str() : nullptr;
}

extern "C" int avsd_plan_bundle_find_plan(const avsd_plan_bundle* b, const char* name) {
  if (!b || !name) return -1;
  for (size_t k = 0; k < b->plans.size(); ++k)
    if (b->plans[k].name == name) return (int)k;
  return -1;
}

extern "C" int avsd_plan_num_calls(const avsd_plan_bundle* b, int plan) {
  return (b && plan >= 0 && plan < (int)b->plans.size()) ? (int)b->plans[plan].calls.size() : -1;
}

extern "C" int avsd_plan_run(avsd_plan_bundle* b, int plan, void* stream) {
  AVSD_REQUIRE(b && plan >= 0 && plan < (int)b->plans.size(), "plan_run: bad plan index");
  Plan& p = b->plans[plan];
  if (!p.resolved) {
    const int rc = resolve(*b, p);
    if (rc != AVSD_OK) return rc;
  }
  for (Call& c : p.calls) {
    for (size_t k = 0; k < c.args.size(); ++k)
      if (c.args[k].tag == 'S') c.resolved[k].p = stream;
    const int rc = c.thunk(c.resolved.data(), (int)c.resolved.size());
    if (rc != AVSD_OK) return rc;   // the entry point has set the error text
  }
  return AVSD_OK;
}

// ---- the operation-level surface SURVEY 8(b)-3 proposed, on top of a bundle recorded with the conventional region names
//      (tools/export_plan.py): "text", "audio" -> plan "set_conditioning";  "x", "t" -> plan "forward" -> "noise_pred";
//      "latents" -> plan "decode" -> "frames" -----------------------------------------------------------------------------
namespace {

int region_ptr(avsd_plan_bundle* b, const char* name, int kind, unsigned char** ptr, int64_t* bytes) {
  const int j = avsd_plan_bundle_find_region(b, name);
  AVSD_REQUIRE(j >= 0, "plan: the bundle has no region '%s' (record it with the conventional names, tools/export_plan.py)", name);
  const Region& r = b->regions[j];
  AVSD_REQUIRE(kind == 0 || r.kind == kind, "plan: region '%s' has kind %d, expected %d", name, r.kind, kind);
  AVSD_REQUIRE(b->buffers[r.buf].ptr != nullptr, "plan: buffer %d (holding region '%s') is not bound", r.buf, name);
  *ptr = static_cast<unsigned char*>(b->buffers[r.buf].ptr) + r.off;
  *bytes = r.bytes;
  return AVSD_OK;
}

int run_named(avsd_plan_bundle* b, const char* plan, void* stream) {
  const int k = avsd_plan_bundle_find_plan(b, plan);
  AVSD_REQUIRE(k >= 0, "plan: the bundle has no plan '%s'", plan);
  return avsd_plan_run(b, k, stream);
}

int copy_in(avsd_plan_bundle* b, const char* name, const void* src, void* stream) {
  unsigned char* dst;
  int64_t bytes;
  const int rc = region_ptr(b, name, AVSD_REGION_INPUT, &dst, &bytes);
  if (rc != AVSD_OK) return rc;
  return src == dst ? AVSD_OK : avsd_copy(src, dst, bytes, 1, stream);
}

}  // namespace

extern "C" int avsd_unet_set_conditioning(avsd_plan_bundle* b, const void* text, const void* audio, void* stream) {
  AVSD_REQUIRE(b && text && audio, "unet_set_conditioning: null argument");
  int rc = copy_in(b, "text", text, stream);
  if (rc == AVSD_OK) rc = copy_in(b, "audio", audio, stream);
  if (rc == AVSD_OK) rc = run_named(b, "set_conditioning", stream);
  return rc;
}

extern "C" int avsd_unet_forward(avsd_plan_bundle* b, const float* sample, const float* timestep, float* noise_pred, void* stream) {
  AVSD_REQUIRE(b && sample && timestep, "unet_forward: null argument");
  int rc = copy_in(b, "x", sample, stream);
  if (rc != AVSD_OK) return rc;
  unsigned char* t;
  int64_t tb;
  rc = region_ptr(b, "t", AVSD_REGION_INPUT, &t, &tb);
  if (rc != AVSD_OK) return rc;
  if (reinterpret_cast<const float*>(t) != timestep) {   // 4 bytes: below avsd_copy's 16-byte granule
    const hipError_t e = hipMemcpyAsync(t, timestep, 4, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream));
    AVSD_REQUIRE(e == hipSuccess, "unet_forward: timestep copy: %s", hipGetErrorString(e));
  }
  rc = run_named(b, "forward", stream);
  if (rc != AVSD_OK || !noise_pred) return rc;
  unsigned char* np;
  int64_t nb;
  rc = region_ptr(b, "noise_pred", AVSD_REGION_OUTPUT, &np, &nb);
  if (rc != AVSD_OK) return rc;
  return avsd_copy(np, noise_pred, nb, 1, stream);
}

extern "C" int avsd_vae_decode(avsd_plan_bundle* b, const float* latents, void* frames_u8, void* stream) {
  AVSD_REQUIRE(b && latents, "vae_decode: null argument");
  int rc = copy_in(b, "latents", latents, stream);
  if (rc == AVSD_OK) rc = run_named(b, "decode", stream);
  if (rc != AVSD_OK || !frames_u8) return rc;
  unsigned char* fp;
  int64_t fb;
  rc = region_ptr(b, "frames", AVSD_REGION_OUTPUT, &fp, &fb);
  if (rc != AVSD_OK) return rc;
  return avsd_copy(fp, frames_u8, fb, 1, stream);
}

extern "C" const void* avsd_plan_region_ptr(avsd_plan_bundle* b, const char* name, int64_t* bytes) {
  unsigned char* p = nullptr;
  int64_t nb = 0;
  if (!b || !name || region_ptr(b, name, 0, &p, &nb) != AVSD_OK) return nullptr;
  if (bytes) *bytes = nb;
  return p;
}
