// cssm_rtc.cpp -- k_propagate specialised for a handle's model structure AT RUN TIME (hipRTC).
//
// The reference composes its models at compile time (Model.scala:110-136: `|+|` builds a new Model value whose f, sde and
// dataLikelihood are closures over the two operands); here a model's structure is data -- one byte per latent component
// (ModelK::comp) -- and the fused kernel branches on it, unless the structure words are template arguments (MKW..): then the
// branches fold away (round 3: -5 % at N = 2^20, -8.5 % at 2^24 on the bench model, -12 % on the LGCP kernel).  Round 3 held five
// such instantiations, the structures of BASELINE's configurations.  This unit gives EVERY model its own: the instantiation
// k_propagate_self / k_propagate_shard / k_propagate<LGCP> <D, IT, OBS, ..., words> of the handle's structure -- with the observation
// model a template argument too, also for the densities the ahead-of-time build keeps behind a runtime switch -- is compiled
// from the library's own kernel sources (embedded at build time: build/rtc_sources.h) with the flags of the ahead-of-time
// build (-O3 -ffp-contract=off: the same arithmetic, the same bits), ~1 s per kernel, cached in memory per process and on disk
// per (sources, flags, instantiation, architecture).  No silent change of kernel: if the runtime compiler is missing or fails,
// one line on stderr says so and the structure-as-data kernel runs.  CSSM_RTC=0 switches the specialisation off.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "cssm_host.h"
#include "build/rtc_sources.h"   // cssm_rtc_sources[]: {include name, text, length} of the kernel headers (Makefile)

namespace {

struct HipRtc {
  void* lib = nullptr;
  int (*CreateProgram)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*AddNameExpression)(void*, const char*) = nullptr;
  int (*CompileProgram)(void*, int, const char**) = nullptr;
  int (*GetLoweredName)(void*, const char*, const char**) = nullptr;
  int (*GetCodeSize)(void*, size_t*) = nullptr;
  int (*GetCode)(void*, char*) = nullptr;
  int (*GetProgramLogSize)(void*, size_t*) = nullptr;
  int (*GetProgramLog)(void*, char*) = nullptr;
  int (*DestroyProgram)(void**) = nullptr;
  int (*Version)(int*, int*) = nullptr;   // optional
  bool ok = false;
};

HipRtc* hiprtc() {
  static HipRtc api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    for (const char* nm : names) { api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (api.lib) break; }
    if (!api.lib) return;
#define SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name))
    SYM(CreateProgram, "hiprtcCreateProgram"); SYM(AddNameExpression, "hiprtcAddNameExpression"); SYM(CompileProgram, "hiprtcCompileProgram");
    SYM(GetLoweredName, "hiprtcGetLoweredName"); SYM(GetCodeSize, "hiprtcGetCodeSize"); SYM(GetCode, "hiprtcGetCode");
    SYM(GetProgramLogSize, "hiprtcGetProgramLogSize"); SYM(GetProgramLog, "hiprtcGetProgramLog"); SYM(DestroyProgram, "hiprtcDestroyProgram"); SYM(Version, "hiprtcVersion");
#undef SYM
    api.ok = api.CreateProgram && api.AddNameExpression && api.CompileProgram && api.GetLoweredName && api.GetCodeSize && api.GetCode && api.DestroyProgram;
  });
  return api.ok ? &api : nullptr;
}

uint64_t fnv1a(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char* b = static_cast<const unsigned char*>(p);
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

// the flags of the ahead-of-time build (Makefile: CXXFLAGS; -DCSSM_RTC_ARCH / -DCSSM_RTC_EXTRA carry its ARCH and EXTRA here)
#ifndef CSSM_RTC_ARCH
#define CSSM_RTC_ARCH "gfx950"
#endif
#ifndef CSSM_RTC_EXTRA
#define CSSM_RTC_EXTRA ""
#endif
std::vector<std::string> build_flags() {
  std::vector<std::string> f = {std::string("--offload-arch=") + CSSM_RTC_ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-mfma"};
  std::string extra = CSSM_RTC_EXTRA;          // (space-separated, as make passes it)
  size_t p = 0;
  while (p < extra.size()) {
    const size_t q = extra.find(' ', p);
    const std::string tok = extra.substr(p, q == std::string::npos ? std::string::npos : q - p);
    if (!tok.empty()) f.push_back(tok);
    if (q == std::string::npos) break;
    p = q + 1;
  }
  return f;
}
const char kEntry[] = "#include \"cssm_propagate.hip.h\"\n";

// what a cached code object depends on: the kernel sources, the flags, the run-time compiler's version (a new ROCm must not be
// handed an old compiler's code) -- the architecture is part of the flags
uint64_t sources_hash() {
  static const uint64_t h = [] {
    uint64_t x = fnv1a(kEntry, sizeof kEntry);
    for (int i = 0; i < cssm_rtc_nsources; ++i) x = fnv1a(cssm_rtc_sources[i].text, cssm_rtc_sources[i].len, x);
    for (const std::string& f : build_flags()) x = fnv1a(f.data(), f.size(), x);
    int major = 0, minor = 0;
    if (HipRtc* r = hiprtc()) if (r->Version) (void)r->Version(&major, &minor);
    const int ver[2] = {major, minor};
    x = fnv1a(ver, sizeof ver, x);
    return x;
  }();
  return h;
}

// where compiled code objects are kept between processes: $CSSM_RTC_CACHE, else rtc_cache/ next to the library (it travels with
// an in-tree build), else a per-user directory under /tmp
std::string cache_dir() {
  static const std::string dir = [] {
    std::vector<std::string> cand;
    if (const char* e = getenv("CSSM_RTC_CACHE")) cand.push_back(e);
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&fnv1a), &info) && info.dli_fname) {
      std::string p = info.dli_fname;
      const size_t s = p.rfind('/');
      cand.push_back((s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/rtc_cache");
    }
    if (const char* x = getenv("XDG_CACHE_HOME")) { if (x[0] == '/') cand.push_back(std::string(x) + "/cssm_rtc_cache"); }
    cand.push_back("/tmp/cssm_rtc_cache_" + std::to_string((unsigned)getuid()));
    // A directory is used only if it is a REAL directory (no symlink), owned by this user and writable by nobody else: the code
    // objects in it are loaded and run.  (A directory somebody else created under the predictable /tmp name is passed over.)
    for (const std::string& d : cand) {
      (void)mkdir(d.c_str(), 0700);
      struct stat sb;
      if (lstat(d.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode) || sb.st_uid != getuid() || (sb.st_mode & (S_IWGRP | S_IWOTH)) != 0) continue;
      if (access(d.c_str(), W_OK | X_OK) == 0) return d;
    }
    return std::string();
  }();
  return dir;
}

struct Entry { hipFunction_t fn = nullptr; bool failed = false; };
std::mutex g_mu;
std::map<std::string, Entry> g_cache;            // key: device:expr
std::atomic<uint64_t> g_compiled{0}, g_disk{0}, g_launches{0}, g_failures{0};

void warn_once(const std::string& expr, const std::string& why) {
  fprintf(stderr, "cssm_pf: structure specialisation of %s unavailable (%s): running the structure-as-data kernel\n", expr.c_str(), why.c_str());
}

bool read_file(const std::string& path, std::vector<char>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  const bool ok = n > 0 && fread(out.data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok;
}

// compile `expr` (a template-id of one of cssm_propagate.hip.h's kernels) -> code object + lowered name
bool compile(const std::string& expr, std::vector<char>& code, std::string& lowered, std::string& why) {
  HipRtc* r = hiprtc();
  if (!r) { why = "libhiprtc.so could not be loaded"; return false; }
  std::vector<const char*> texts, names;
  for (int i = 0; i < cssm_rtc_nsources; ++i) { texts.push_back(cssm_rtc_sources[i].text); names.push_back(cssm_rtc_sources[i].name); }
  void* prog = nullptr;
  if (r->CreateProgram(&prog, kEntry, "cssm_rtc.hip", (int)texts.size(), texts.data(), names.data()) != 0) { why = "hiprtcCreateProgram failed"; return false; }
  bool ok = r->AddNameExpression(prog, expr.c_str()) == 0;
  if (!ok) why = "hiprtcAddNameExpression failed";
  const std::vector<std::string> flags = build_flags();
  std::vector<const char*> fl;
  for (const std::string& f : flags) fl.push_back(f.c_str());
  if (ok && r->CompileProgram(prog, (int)fl.size(), fl.data()) != 0) {
    ok = false;
    size_t n = 0;
    std::string log;
    if (r->GetProgramLogSize && r->GetProgramLog && r->GetProgramLogSize(prog, &n) == 0 && n > 1) { log.resize(n); r->GetProgramLog(prog, &log[0]); }
    why = "hiprtcCompileProgram failed: " + log.substr(0, 600);
  }
  if (ok) {
    const char* low = nullptr;
    size_t n = 0;
    ok = r->GetLoweredName(prog, expr.c_str(), &low) == 0 && low && r->GetCodeSize(prog, &n) == 0 && n > 0;
    if (ok) { lowered = low; code.resize(n); ok = r->GetCode(prog, code.data()) == 0; }
    if (!ok) why = "no code object for the instantiation";
  }
  r->DestroyProgram(&prog);
  return ok;
}

}  // namespace

// The kernel `expr` for the current device: from the process cache, the disk cache, or the runtime compiler.  nullptr (after one line
// on stderr per expression): the caller launches the structure-as-data kernel.
hipFunction_t cssm_rtc_function(const std::string& expr) {
  static const bool off = [] { const char* e = getenv("CSSM_RTC"); return e && e[0] == '0'; }();
  if (off) return nullptr;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const std::string key = std::to_string(dev) + ":" + expr;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_cache.find(key);
  if (it != g_cache.end()) return it->second.failed ? nullptr : it->second.fn;
  Entry e;
  std::string why, lowered;
  std::vector<char> code;
  char tag[64];
  snprintf(tag, sizeof tag, "%016llx_%016llx", (unsigned long long)sources_hash(), (unsigned long long)fnv1a(expr.data(), expr.size()));
  const std::string dir = cache_dir();
  const std::string base = dir.empty() ? std::string() : dir + "/" + tag;
  bool have = false;
  if (!base.empty()) {   // <tag>.hsaco + <tag>.name (the lowered name of the instantiation)
    std::vector<char> nm;
    if (read_file(base + ".hsaco", code) && read_file(base + ".name", nm)) { lowered.assign(nm.begin(), nm.end()); have = true; g_disk++; }
  }
  if (!have) {
    if (!compile(expr, code, lowered, why)) { e.failed = true; g_failures++; warn_once(expr, why); g_cache[key] = e; return nullptr; }
    g_compiled++;
    if (!base.empty()) {   // written under a temporary name and renamed: a reader never sees half a file
      const std::string tmp = base + ".tmp" + std::to_string((long)getpid());
      FILE* f = fopen(tmp.c_str(), "wb");
      if (f) {
        const bool w = fwrite(code.data(), 1, code.size(), f) == code.size();
        fclose(f);
        // the lowered name first, under a temporary name and renamed like the code object: a concurrent rank finds either both
        // files whole or no .hsaco at all (it looks for the .hsaco first) -- never half a name
        const std::string tmpn = base + ".ntmp" + std::to_string((long)getpid());
        FILE* g = w ? fopen(tmpn.c_str(), "wb") : nullptr;
        const bool wn = g && fwrite(lowered.data(), 1, lowered.size(), g) == lowered.size();
        if (g) fclose(g);
        if (!(wn && rename(tmpn.c_str(), (base + ".name").c_str()) == 0 && rename(tmp.c_str(), (base + ".hsaco").c_str()) == 0)) {
          (void)remove(tmpn.c_str());
          (void)remove(tmp.c_str());
        }
      }
    }
  }
  hipModule_t mod = nullptr;
  hipError_t he = hipModuleLoadData(&mod, code.data());
  if (he == hipSuccess) he = hipModuleGetFunction(&e.fn, mod, lowered.c_str());
  if (he != hipSuccess) {
    (void)hipGetLastError();
    e.failed = true; e.fn = nullptr; g_failures++;
    warn_once(expr, std::string("loading the code object: ") + hipGetErrorString(he));
  }
  g_cache[key] = e;
  return e.failed ? nullptr : e.fn;
}

// The fused kernel of launch `a` specialised for the handle's structure words and observation model; false: not launched
// (specialisation off, latent dimension beyond the twelve components three words cover, or the runtime compiler unavailable).
// kind: 0 = k_propagate_self<D, IT, OB, SUMS, ONEV, W..> (SUMS = a.sums ? 1 : 0), 1 = k_propagate_shard<D, IT, OB, ONEV, W..>,
// 2 = k_propagate<D, true, IT, -1, SM, W..> (LGCP; SM = 0 / 1 / 2 as the dispatcher of cssm_prop.hip chooses),
// 3 = k_propagate_batch<D, IT, OB, ONEV, W..> (B chains per launch: cssm_batch.hip).
bool cssm_rtc_launch(const PropLaunch& a, int kind, int D, int IT, int onev) {
  if (!a.specialise || D > 12 || a.mk.d != D) return false;
  char expr[256];
  const unsigned w0 = a.mk.comp[0], w1 = a.mk.comp[1], w2 = a.mk.comp[2];
  const int ob = a.obs_kind;      // the observation model at compile time, whichever it is
  if (kind == 0) snprintf(expr, sizeof expr, "k_propagate_self<%d, %d, %d, %d, %d, %uu, %uu, %uu>", D, IT, ob, a.sums ? 1 : 0, onev, w0, w1, w2);
  else if (kind == 1) snprintf(expr, sizeof expr, "k_propagate_shard<%d, %d, %d, %d, %uu, %uu, %uu>", D, IT, ob, onev, w0, w1, w2);
  else if (kind == 3) snprintf(expr, sizeof expr, "k_propagate_batch<%d, %d, %d, %d, %uu, %uu, %uu>", D, IT, ob, onev, w0, w1, w2);
  else snprintf(expr, sizeof expr, "k_propagate<%d, true, %d, -1, %d, %uu, %uu, %uu>", D, IT, !a.sums ? 0 : (a.sharded ? 2 : 1), w0, w1, w2);
  hipFunction_t fn = cssm_rtc_function(expr);
  if (!fn) return false;
  // (copies: the argument array points at them)
  const double* src = a.src; size_t src_stride = a.src_stride; const uint32_t* anc = a.anc; double* dst = a.dst; size_t dst_stride = a.dst_stride;
  double* logw = a.logw; uint64_t n = a.n, gid0 = a.gid0, seed = a.seed; const StepRec* rec = a.rec; ModelK mk = a.mk; Scalars* sc = a.sc;
  int slot_set = a.slot_set; const double* src2 = a.src2; size_t src2_stride = a.src2_stride; uint32_t n_split = a.n_split;
  const double* logtab = a.logtab; uint64_t chunk = a.chunk; int do_sums = a.do_sums; cssm_u128* subS = a.subS; cssm_u128* subS2 = a.subS2;
  double* pick_out = a.pick_out; uint32_t pick_slot = a.pick_slot, step = a.step; const double* fsub = a.fsub;
  const void* chains = a.chains; int cur = a.cur, anc_valid = a.anc_valid, want_pick = a.want_pick; uint32_t rec_idx = a.rec_idx;
  unsigned grid_y = 1;
  std::vector<void*> args;
  if (kind == 3) {
    args = {&chains, &cur, &anc_valid, &src_stride, &n, &rec_idx, &mk, &slot_set, &logtab, &chunk, &want_pick, &step};
    grid_y = (unsigned)a.nchains;
  } else if (kind == 0)
    args = {&src, &src_stride, &anc, &dst, &dst_stride, &logw, &n, &seed, &rec, &mk, &sc, &slot_set, &logtab, &chunk, &subS, &subS2, &pick_out, &pick_slot, &step};
  else if (kind == 1)
    args = {&src, &src_stride, &anc, &dst, &dst_stride, &logw, &n, &gid0, &seed, &rec, &mk, &sc, &src2, &n_split, &logtab, &chunk, &subS, &subS2, &step, &slot_set};
  else
    args = {&src, &src_stride, &anc, &dst, &dst_stride, &logw, &n, &gid0, &seed, &rec, &mk, &sc, &slot_set, &src2, &src2_stride, &n_split, &logtab, &chunk,
            &do_sums, &subS, &subS2, &pick_out, &pick_slot, &fsub};
  if (hipModuleLaunchKernel(fn, (unsigned)a.grid, grid_y, 1, CSSM_BLOCK, 1, 1, 0, a.stream, args.data(), nullptr) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  g_launches++;
  return true;
}

extern "C" int cssm_rtc_info(uint64_t* out4) {
  if (!out4) return CSSM_EINVAL_ARG;
  out4[0] = g_compiled; out4[1] = g_disk; out4[2] = g_launches; out4[3] = g_failures;
  return CSSM_OK;
}
