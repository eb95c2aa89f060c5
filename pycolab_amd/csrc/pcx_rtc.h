// pcx_rtc.h -- host side only: what the kernels that are ALSO built at run time share (pcx_generic_step since round 4,
// pcx_scrolly_maze_step since round 6): hiprtc resolved with dlopen (part of the ROCm runtime; no link dependency), and the
// cache of code objects -- in the process per (constants, device), on disk under $PCX_JIT_CACHE (default: jit_cache/ next to
// libpcx.so, else the user's cache directory), keyed by a hash of everything that goes into a build: the embedded sources,
// the options, the constants and the compiler's version.  Nothing here decides what a failed build means: the callers fall
// back to the instances libpcx.so was built with (same results).  gfx950 only.
#pragma once

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

namespace pcx {
namespace rtc {

struct Api {
  void* lib = nullptr;
  int (*create)(void**, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
  int (*compile)(void*, int, const char* const*) = nullptr;
  int (*log_size)(void*, size_t*) = nullptr;
  int (*log)(void*, char*) = nullptr;
  int (*code_size)(void*, size_t*) = nullptr;
  int (*code)(void*, char*) = nullptr;
  int (*destroy)(void**) = nullptr;
  int (*version)(int*, int*) = nullptr;
  bool ok = false;
};

inline const Api& api() {
  static Api a = [] {
    Api x;
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (x.lib) break;
    }
    if (!x.lib) return x;
    auto sym = [&](const char* n) { return dlsym(x.lib, n); };
    x.create = reinterpret_cast<decltype(x.create)>(sym("hiprtcCreateProgram"));
    x.compile = reinterpret_cast<decltype(x.compile)>(sym("hiprtcCompileProgram"));
    x.log_size = reinterpret_cast<decltype(x.log_size)>(sym("hiprtcGetProgramLogSize"));
    x.log = reinterpret_cast<decltype(x.log)>(sym("hiprtcGetProgramLog"));
    x.code_size = reinterpret_cast<decltype(x.code_size)>(sym("hiprtcGetCodeSize"));
    x.code = reinterpret_cast<decltype(x.code)>(sym("hiprtcGetCode"));
    x.destroy = reinterpret_cast<decltype(x.destroy)>(sym("hiprtcDestroyProgram"));
    x.version = reinterpret_cast<decltype(x.version)>(sym("hiprtcVersion"));
    x.ok = x.create && x.compile && x.log_size && x.log && x.code_size && x.code && x.destroy;
    return x;
  }();
  return a;
}

inline uint64_t fnv(uint64_t h, const void* p, size_t n) {
  const unsigned char* b = static_cast<const unsigned char*>(p);
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001B3ull; }
  return h;
}

// Where the code objects are kept: $PCX_JIT_CACHE, else jit_cache/ next to libpcx.so -- and when that directory cannot
// be written (a package installed read-only), $XDG_CACHE_HOME/pcx_jit or ~/.cache/pcx_jit, so that such a host compiles a
// template once and not in every process.  `for_write`: the directory a new entry goes to (created if need be).
inline std::string lib_cache_dir() {
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&lib_cache_dir), &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t slash = p.rfind('/');
    return (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/jit_cache";
  }
  return "";
}
inline std::string user_cache_dir() {
  if (const char* x = getenv("XDG_CACHE_HOME")) if (*x) return std::string(x) + "/pcx_jit";
  if (const char* h = getenv("HOME")) if (*h) return std::string(h) + "/.cache/pcx_jit";
  return "";
}
inline bool writable_dir(const std::string& d) {
  if (d.empty()) return false;
  // every missing directory on the way (~/.cache may not exist yet: a single mkdir left the host recompiling in every process)
  for (size_t i = 1; i <= d.size(); ++i)
    if (i == d.size() || d[i] == '/') mkdir(d.substr(0, i).c_str(), 0755);
  return access(d.c_str(), W_OK | X_OK) == 0;
}
inline std::vector<std::string> cache_dirs() {  // read order; the first writable one takes new entries
  if (const char* e = getenv("PCX_JIT_CACHE")) return {e};
  std::vector<std::string> v;
  const std::string a = lib_cache_dir(), b = user_cache_dir();
  if (!a.empty()) v.push_back(a);
  if (!b.empty()) v.push_back(b);
  return v;
}

// One run-time build: the embedded headers (tools/embed_sources.py: names / sources), the one-line translation unit that
// includes the kernel's header, the options (one of them names the header of constants, `spec_name`, which code_object()
// adds to the program), and what the cache files and the messages are called.
struct Program {
  const char* const* names;
  const char* const* sources;
  int count;
  const char* tu;        // e.g. "#include \"pcx_generic_kernel.h\"\n"
  const char* tu_name;
  const char* const* options;
  int n_options;
  const char* spec_name;  // the include name of the constants' header
  const char* prefix;     // cache files: <prefix>_<hash>.hsaco
  const char* tag;        // messages: "[<tag>] ..."
};

// The code object of program `p` for the constants in `spec` (the text of the header p.spec_name): from the disk cache, or
// compiled now.  Empty + `why` on failure.
// read_cache false: compile even if the cache has an entry (and replace it) -- what load() asks for when the entry it
// got would not load (a file damaged on disk must not pin the engine to the table-driven build for good).
inline std::vector<char> code_object(const Program& p, const std::string& spec, std::string& why, bool verbose, bool read_cache = true, bool* from_cache = nullptr) {
  if (from_cache) *from_cache = false;
  uint64_t h = 0xCBF29CE484222325ull;
  for (int i = 0; i < p.count; ++i) h = fnv(h, p.sources[i], strlen(p.sources[i]) + 1);
  h = fnv(h, p.tu, strlen(p.tu) + 1);
  for (int i = 0; i < p.n_options; ++i) h = fnv(h, p.options[i], strlen(p.options[i]) + 1);
  h = fnv(h, spec.data(), spec.size());
  // the COMPILER only: hiprtc's own version where the library is there, the HIP version this file was built with where it
  // is not -- so that a cache filled by compiler.prebuild() on a build box is hit on a host without libhiprtc.  (Round 5
  // also hashed hipRuntimeGetVersion(): a patch-level difference between the two boxes then missed the cache on exactly the
  // host that cannot recompile; ADVICE r5.  A code object depends on what compiled it, not on the runtime that loads it.)
  const Api& r = api();
  int vmaj = HIP_VERSION_MAJOR, vmin = HIP_VERSION_MINOR;
  if (r.ok && r.version) r.version(&vmaj, &vmin);
  h = fnv(h, &vmaj, sizeof vmaj); h = fnv(h, &vmin, sizeof vmin);
  char name[96];
  snprintf(name, sizeof name, "/%s_%016llx.hsaco", p.prefix, (unsigned long long)h);
  const std::vector<std::string> dirs = cache_dirs();
  std::vector<char> code;
  for (const std::string& dir : dirs) {
    if (!read_cache) break;
    const std::string path = dir + name;
    if (FILE* f = fopen(path.c_str(), "rb")) {
      fseek(f, 0, SEEK_END);
      const long n = ftell(f);
      fseek(f, 0, SEEK_SET);
      if (n > 0) { code.resize((size_t)n); if (fread(code.data(), 1, (size_t)n, f) != (size_t)n) code.clear(); }
      fclose(f);
      if (!code.empty()) {
        if (verbose) fprintf(stderr, "[%s] specialised kernel from %s\n", p.tag, path.c_str());
        if (from_cache) *from_cache = true;
        return code;
      }
    }
  }
  if (!r.ok) { why = "libhiprtc.so is not available"; return code; }
  std::vector<const char*> names(p.names, p.names + p.count), sources(p.sources, p.sources + p.count);
  names.push_back(p.spec_name);
  sources.push_back(spec.c_str());
  void* prog = nullptr;
  if (r.create(&prog, p.tu, p.tu_name, (int)names.size(), sources.data(), names.data()) != 0) { why = "hiprtcCreateProgram failed"; return code; }
  const int rc = r.compile(prog, p.n_options, p.options);
  if (rc != 0) {
    size_t n = 0;
    r.log_size(prog, &n);
    std::string log(n, '\0');
    if (n) r.log(prog, &log[0]);
    why = "hiprtcCompileProgram failed: " + log.substr(0, 2000);
    r.destroy(&prog);
    return code;
  }
  size_t n = 0;
  r.code_size(prog, &n);
  code.resize(n);
  if (n) r.code(prog, code.data());
  r.destroy(&prog);
  if (code.empty()) { why = "hiprtc produced no code"; return code; }
  for (const std::string& dir : dirs) {  // best effort: a private temporary file (mkstemp: threads and processes may race for
    if (!writable_dir(dir)) continue;     // the same entry), closed with its errors checked, then renamed into place
    std::string tpath = dir + name + ".XXXXXX";
    const int fd = mkstemp(&tpath[0]);
    if (fd < 0) continue;
    FILE* f = fdopen(fd, "wb");
    if (!f) { close(fd); unlink(tpath.c_str()); continue; }
    const bool wrote = fwrite(code.data(), 1, code.size(), f) == code.size();
    const bool closed = fclose(f) == 0;
    const std::string path = dir + name;
    if (!wrote || !closed || rename(tpath.c_str(), path.c_str()) != 0) { unlink(tpath.c_str()); continue; }
    chmod(path.c_str(), 0644);
    break;
  }
  if (verbose) fprintf(stderr, "[%s] specialised kernel compiled (%zu bytes)\n", p.tag, code.size());
  return code;
}

}  // namespace rtc
}  // namespace pcx
