// smpl_amd/csrc/specialize.cpp -- see specialize.h
#include "specialize.h"

#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "rtc_compile.h"
#include "kernels.h"
#include "model_compile.h"

extern char** environ;

namespace smplx {

namespace {

const char* const kNames[K_COUNT] = {"k_state_prep", "k_expand", "k_pipe_prep", "k_pipe_setup", "k_pipe_configs",
                                     "k_pipe_finish", "k_small_batch", "k_edge_valid", "k_state_valid", "k_heuristic",
                                     "k_sphere_positions", "k_search"};

uint64_t fnv1a(uint64_t h, const std::string& s)
{
    for (unsigned char c : s) { h ^= c; h *= 0x100000001B3ull; }
    return h;
}

std::string cache_dir()
{
    std::string d;
    if (const char* e = getenv("SMPLX_CACHE_DIR")) d = e;
    else if (const char* x = getenv("XDG_CACHE_HOME")) d = std::string(x) + "/smpl_amd";
    else if (const char* h = getenv("HOME")) d = std::string(h) + "/.cache/smpl_amd";
    else d = "/tmp/smpl_amd_cache";
    // create the last two levels; failure just disables the disk cache
    const size_t cut = d.find_last_of('/');
    if (cut != std::string::npos && cut > 0) (void)mkdir(d.substr(0, cut).c_str(), 0755);
    (void)mkdir(d.c_str(), 0755);
    return d;
}

bool read_file(const std::string& path, std::vector<char>& out)
{
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

void write_file_atomic(const std::string& path, const std::vector<char>& data)
{
    const std::string tmp = path + "." + std::to_string((long)getpid()) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
    fclose(f);
    if (ok) (void)rename(tmp.c_str(), path.c_str());
    else (void)remove(tmp.c_str());
}

bool compile_via_helper(const std::string& header, const std::string& dir, uint64_t h, std::vector<char>& code, std::string& why);

// Runs smplx_rtc (next to the shared library) as a child process: header file in, code object file out.
bool compile_via_helper(const std::string& header, const std::string& dir, uint64_t h, std::vector<char>& code, std::string& why)
{
    Dl_info info;
    if (!dladdr((const void*)&compile_via_helper, &info) || !info.dli_fname) { why = "dladdr failed"; return false; }
    std::string exe = info.dli_fname;
    const size_t cut = exe.find_last_of('/');
    exe = (cut == std::string::npos ? std::string(".") : exe.substr(0, cut)) + "/smplx_rtc";
    if (access(exe.c_str(), X_OK) != 0) { why = exe + " not found"; return false; }
    char tag[64];
    snprintf(tag, sizeof tag, "/%016llx.%ld", (unsigned long long)h, (long)getpid());
    const std::string hpath = dir + tag + ".h", opath = dir + tag + ".out";
    {
        FILE* f = fopen(hpath.c_str(), "wb");
        if (!f) { why = "cannot write " + hpath; return false; }
        fwrite(header.data(), 1, header.size(), f);
        fclose(f);
    }
    std::vector<std::string> args = {exe, hpath, opath};
    for (const std::string& x : rtc_extra_defines()) args.push_back(x);
    std::vector<char*> argv;
    for (std::string& a : args) argv.push_back(&a[0]);
    argv.push_back(nullptr);
    // the child only compiles: keep tool preloads of the host process (profilers, sanitizers) out of it
    std::vector<char*> envp;
    for (char** e = environ; e && *e; ++e) {
        if (!strncmp(*e, "LD_PRELOAD=", 11) || !strncmp(*e, "HSA_TOOLS_LIB=", 14) || !strncmp(*e, "ROCP", 4) ||
            !strncmp(*e, "ROCPROFILER", 11))
            continue;
        envp.push_back(*e);
    }
    envp.push_back(nullptr);
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, exe.c_str(), nullptr, nullptr, argv.data(), envp.data());
    bool ok = false;
    if (rc != 0) {
        why = std::string("posix_spawn: ") + strerror(rc);
    } else {
        int status = 0;
        while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
        if (WIFEXITED(status) && WEXITSTATUS(status) == 0) ok = read_file(opath, code);
        if (!ok) why = "smplx_rtc failed (status " + std::to_string(status) + ")";
    }
    (void)remove(hpath.c_str());
    (void)remove(opath.c_str());
    return ok;
}

struct Loaded {
    bool ok = false;
    std::string why;   // failure reason, or on success where the code object came from
    hipFunction_t fn[K_COUNT] = {};
};

std::mutex g_mutex;
std::map<std::pair<uint64_t, int>, Loaded> g_loaded;   // (model hash, device) -> module functions; never unloaded

}  // namespace

void generic_kernels(KernelSet& ks)
{
    const void* g[K_COUNT] = {(const void*)k_state_prep, (const void*)k_expand, (const void*)k_pipe_prep,
                              (const void*)k_pipe_setup, (const void*)k_pipe_configs, (const void*)k_pipe_finish,
                              (const void*)k_small_batch, (const void*)k_edge_valid, (const void*)k_state_valid,
                              (const void*)k_heuristic, (const void*)k_sphere_positions, (const void*)k_search};
    for (int i = 0; i < K_COUNT; ++i) { ks.k[i].fn = nullptr; ks.k[i].generic = g[i]; }
    ks.specialized = false;
}

bool specialized_kernels(const SmplxModelDev& model, KernelSet& ks, std::string& why)
{
    generic_kernels(ks);
    const std::string header = model_const_header(model);
    uint64_t h = 0xCBF29CE484222325ull;
    h = fnv1a(h, header);
    h = fnv1a(h, SRC_KERNELS_HIP);
    h = fnv1a(h, SRC_DET_MATH_H);
    h = fnv1a(h, SRC_DEVICE_TYPES_H);
    h = fnv1a(h, SRC_KERNELS_H);
    h = fnv1a(h, SRC_SEARCH_KERNEL_H);
    for (const char* o : kRtcOptions) h = fnv1a(h, o);
    for (const std::string& x : rtc_extra_defines()) h = fnv1a(h, x);
    int major = 0, minor = 0;
    (void)hiprtcVersion(&major, &minor);
    h = fnv1a(h, std::to_string(major) + "." + std::to_string(minor));
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { why = "no current device"; return false; }

    std::lock_guard<std::mutex> lock(g_mutex);
    Loaded& L = g_loaded[{h, dev}];
    if (!L.ok && L.why.empty()) {   // first request for this model on this device
        char name[64];
        snprintf(name, sizeof name, "/%016llx.hsaco", (unsigned long long)h);
        const std::string path = cache_dir() + name;
        std::vector<char> code;
        bool have = read_file(path, code);
        std::string how = "disk cache";
        if (!have) {
            // the helper executable compiles with the ROCm the library was built against; a host program may have
            // loaded another hiprtc/comgr first (PyTorch bundles its own), whose code generation differs
            std::string why_helper;
            have = compile_via_helper(header, cache_dir(), h, code, why_helper);
            how = "compiled by smplx_rtc";
            if (have) {
                write_file_atomic(path, code);
            } else {
                // no helper: whatever hiprtc this process has; cached under another name so that a later run with
                // the helper does not pick it up
                const std::string path2 = path.substr(0, path.size() - 6) + ".inproc.hsaco";
                have = read_file(path2, code);
                how = "disk cache (in-process build)";
                if (!have) {
                    have = rtc_compile(header, code, L.why);
                    how = "compiled in-process (" + why_helper + ")";
                    if (have) write_file_atomic(path2, code);
                    else L.why = "helper: " + why_helper + "; in-process: " + L.why;
                }
            }
        }
        if (have) {
            hipModule_t mod = nullptr;
            hipError_t e = hipModuleLoadData(&mod, code.data());
            if (e != hipSuccess) {
                L.why = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
            } else {
                L.ok = true;
                for (int i = 0; i < K_COUNT && L.ok; ++i) {
                    e = hipModuleGetFunction(&L.fn[i], mod, kNames[i]);
                    if (e != hipSuccess) { L.ok = false; L.why = std::string("hipModuleGetFunction ") + kNames[i] + ": " + hipGetErrorString(e); }
                }
            }
        }
        if (!L.ok && L.why.empty()) L.why = "specialisation failed";
        if (L.ok) L.why = how;
    }
    why = L.why;
    if (!L.ok) return false;
    for (int i = 0; i < K_COUNT; ++i) ks.k[i].fn = L.fn[i];
    ks.specialized = true;
    return true;
}

}  // namespace smplx
