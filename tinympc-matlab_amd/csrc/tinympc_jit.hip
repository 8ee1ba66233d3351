// tinympc_jit.hip -- run-time specialisation of the layout-D solve kernels (host only).
//
// Layout D (tinympc_solve_d.hip, _dw.hip, _dx.hip for 16 / 32 / 64 lanes per instance) makes the horizon and the system
// size compile-time constants: that is what lets every knot's dual live in its own register pair and two wavefronts share
// a SIMD. The library carries a handful of instantiations (BASELINE shapes); every OTHER shape that fits the register / LDS
// plan gets the same kernel through hiprtc: on first use the very same source file is compiled with
// -DTINY_JIT -DTINY_JIT_NX=.. -DTINY_JIT_NU=.. -DTINY_JIT_N=.. -DTINY_JIT_VREG=.. (one `extern "C"` kernel, static LDS), the
// code object is cached in memory per device and on disk (~/.cache/tinympc_hip or $TINYMPC_JIT_CACHE), and launched with
// hipModuleLaunchKernel. No hipcc, no host compiler: hiprtc is part of the ROCm runtime. TINYMPC_JIT=0 switches it off; if
// anything fails (no hiprtc, sources not found, compile error) the shape simply keeps running on layout B / A.
//
// The kernels' sources are found next to the library: <dir of libtinympc_hip.so>/csrc and <dir>/../include, or where
// $TINYMPC_HIP_SRC / $TINYMPC_HIP_INCLUDE point.
#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "tinympc_device.h"

namespace tinympc {

namespace {

struct JitPlan {
    bool ok = false;
    int vreg = 0;  // slack knots kept in registers
    int wps = 2;   // wavefronts per SIMD: 2 (256 registers each), or 1 (512 registers) for long horizons
    int wpg = 8;   // wavefronts per workgroup: 4 where the LDS plan allows it (spreads mid-size batches over the CUs), else 8
    size_t lds_bytes = 0;  // per workgroup
    const char *source = nullptr;
};

// The register / LDS plan of tinympc_solve_d*.hip, re-derived here (the kernels static_assert the LDS side):
//   VGPRs  2*(N-1) for the duals + 2*vreg for the register part of the slack + the operator row + ~76 for everything else
//          (+ the families' / adaptive rho's registers) must stay <= 256 (two wavefronts per SIMD) or <= 512 (one);
//   LDS    per wave (N-1-vreg) slack rows of 512 B + the feed-forward d; the workgroup's waves + the operators + tables must fit
//          its share of the CU's 160 KB.
JitPlan plan_for(int W, int nx, int nu, int N, bool ct, bool fam = false, bool adapt = false) {
    JitPlan pl;
    const int nxu = nx + nu, ns = N - 1;
    if (N < 4 || nx < 1 || nu < 1) return pl;
    int mregs, ops_doubles, d_doubles;
    if (W == 16 && nxu <= 16) { mregs = 32; ops_doubles = 2 * 16 * 16; d_doubles = ((ns * 4 * nu) + 1) & ~1; pl.source = "tinympc_solve_d.hip"; }
    else if (W == 32 && nxu > 16 && nxu <= 32) { mregs = 64; ops_doubles = 2 * 32 * 32; d_doubles = ((ns * 2 * nu) + 1) & ~1; pl.source = "tinympc_solve_dw.hip"; }
    else if (W == 64 && nxu > 32 && nxu <= 64) { mregs = 128; ops_doubles = 2 * 64 * 64; d_doubles = ((ns * nu) + 1) & ~1; pl.source = "tinympc_solve_dx.hip"; }
    else return pl;
    if ((fam || adapt) && W != 16) return pl;  // cone / linear families, adaptive rho: 16-lane form only
    if (fam && adapt) return pl;
    if (adapt) ops_doubles += 5 * 16 * 16;  // derivative rows of the two operators, [A'; B'], Pinf and its derivative
    // the workgroup's LDS copy of the per-knot tables (+ the linear rows' coefficients of the families)
    const int tab_doubles = (ct ? 0 : 3 * (N + 2) * W + W) + (fam ? 3 * MAX_LIN_ROWS * 16 : 0);
    // families: three more register pairs per knot (gc, gl, lx), the three mask rows (3 x 16 doubles) and their scalars
    // adaptive rho: the [A'; B'] row during an adaptation sweep, the lane's rho / pNref and the four norms
    const int fam_regs = (fam ? 6 * ns + 96 + 24 : 0) + (adapt ? 32 + 28 : 0);
    // two wavefronts per SIMD first (workgroups of four, else eight); a horizon that does not fit gets one wavefront per SIMD with
    // all 512 registers and a quarter of the CU's LDS -- the chain latency is then exposed, but nothing of the state leaves the chip
    static const int cand[3][2] = {{2, 4}, {2, 8}, {1, 4}};  // (wavefronts per SIMD, per workgroup), in order of preference
    for (const auto &c : cand) {
        const int wps = c[0], wpg = c[1];
        const int budget = 256 * (3 - wps) - (wps == 2 ? 76 : 110) - (ct ? 0 : 8) - mregs - 2 * ns - fam_regs;
        if (budget < 0) continue;
        int vreg = budget / 2;
        if (vreg > ns) vreg = ns;
        // the workgroup's LDS share: wpg of the 4 * wps wavefronts a CU holds
        const int wave_doubles = (160 * 1024 / 8 * wpg / (4 * wps) - ops_doubles - tab_doubles) / wpg - d_doubles;
        if (wave_doubles < 0) continue;
        const int vlmax = wave_doubles / 64;
        if (ns - vreg > vlmax) continue;  // the LDS part of the slack does not fit
        pl.ok = true;
        pl.vreg = vreg;
        pl.wps = wps;
        pl.wpg = wpg;
        pl.lds_bytes = sizeof(double) * ((size_t)ops_doubles + tab_doubles + (size_t)wpg * ((size_t)(ns - vreg) * 64 + d_doubles));
        return pl;
    }
    return pl;
}

bool jit_enabled() {
    const char *e = getenv("TINYMPC_JIT");
    return !(e && e[0] == '0');
}

std::string dir_of_library() {
    Dl_info info;
    if (dladdr(reinterpret_cast<const void *>(&plan_for), &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        const size_t k = p.rfind('/');
        return k == std::string::npos ? std::string(".") : p.substr(0, k);
    }
    return ".";
}

bool file_exists(const std::string &p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

std::string source_dir() {
    if (const char *e = getenv("TINYMPC_HIP_SRC")) return e;
    return dir_of_library() + "/csrc";
}
std::string include_dir() {
    if (const char *e = getenv("TINYMPC_HIP_INCLUDE")) return e;
    return dir_of_library() + "/../include";
}

std::string read_file(const std::string &p) {
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

unsigned long long fnv1a(const std::string &s, unsigned long long h = 1469598103934665603ull) {
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

std::string cache_dir() {
    if (const char *e = getenv("TINYMPC_JIT_CACHE")) return e;
    const char *home = getenv("HOME");
    return std::string(home ? home : "/tmp") + "/.cache/tinympc_hip";
}

struct JitKernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    bool failed = false;
    int wpg = 8;  // wavefronts per workgroup
    std::vector<char> image;  // the code object stays alive as long as the module does
};
using Key = std::tuple<int, int, int, int, int, int>;  // device, W, nx, nu, N, constant tables + 2 * families + 4 * adaptive rho
std::map<Key, JitKernel> &cache() {
    static std::map<Key, JitKernel> c;
    return c;
}
std::mutex &cache_mutex() {
    static std::mutex m;
    return m;
}

// Compile (or fetch from the disk cache) the code object of one shape. Empty on failure; `why` says why.
std::vector<char> build_code_object(const JitPlan &pl, int nx, int nu, int N, bool ct, bool fam, bool adapt, const std::string &arch, bool use_disk_cache, std::string &why) {
    const std::string sdir = source_dir(), idir = include_dir();
    const std::string spath = sdir + "/" + pl.source;
    if (!file_exists(spath) || !file_exists(sdir + "/tinympc_device.h")) {
        why = "kernel sources not found in " + sdir + " (set TINYMPC_HIP_SRC)";
        return {};
    }
    const std::string src = read_file(spath);
    // everything the translation unit reads goes into the cache key
    unsigned long long h = fnv1a(src);
    for (const char *dep : {"tinympc_device.h", "tinympc_sweep.h", "tinympc_solve_d_chain.h", "tinympc_solve_dw_chain.h", "tinympc_solve_dx_chain.h"})
        h = fnv1a(read_file(sdir + "/" + dep), h);
    char shape[160];
    snprintf(shape, sizeof(shape), "%s nx=%d nu=%d N=%d vreg=%d wps=%d wpg=%d ct=%d fam=%d adapt=%d %s", pl.source, nx, nu, N, pl.vreg, pl.wps, pl.wpg, (int)ct, (int)fam, (int)adapt, arch.c_str());
    h = fnv1a(shape, h);
    char name[64];
    snprintf(name, sizeof(name), "/jit_%016llx.hsaco", h);
    const std::string cpath = cache_dir() + name;
    if (use_disk_cache && file_exists(cpath)) {
        const std::string blob = read_file(cpath);
        if (!blob.empty()) return std::vector<char>(blob.begin(), blob.end());
    }
    if (!use_disk_cache) (void)unlink(cpath.c_str());  // (a cached image that did not load: replace it)
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), pl.source, 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        why = "hiprtcCreateProgram failed";
        return {};
    }
    std::vector<std::string> o = {"--offload-arch=" + arch, "-O3", "-std=c++17", "-I" + sdir, "-I" + idir, "-DTINY_JIT=1",
                                  "-DTINY_JIT_NX=" + std::to_string(nx), "-DTINY_JIT_NU=" + std::to_string(nu),
                                  "-DTINY_JIT_N=" + std::to_string(N), "-DTINY_JIT_VREG=" + std::to_string(pl.vreg),
                                  "-DTINY_JIT_WPS=" + std::to_string(pl.wps), "-DTINY_JIT_WPG=" + std::to_string(pl.wpg), std::string("-DTINY_JIT_CT=") + (ct ? "1" : "0"),
                                  std::string("-DTINY_JIT_FAM=") + (fam ? "1" : "0"), std::string("-DTINY_JIT_ADAPT=") + (adapt ? "1" : "0")};
    std::vector<const char *> opts;
    for (const auto &x : o) opts.push_back(x.c_str());
    const hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        why = std::string("hiprtc: ") + hiprtcGetErrorString(r) + "\n" + log.substr(0, 2000);
        hiprtcDestroyProgram(&prog);
        return {};
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    // disk cache, best effort (atomic: write beside, rename)
    const std::string cdir = cache_dir();
    {
        std::string acc;
        for (size_t i = 1; i <= cdir.size(); ++i)
            if (i == cdir.size() || cdir[i] == '/') {
                acc = cdir.substr(0, i);
                (void)mkdir(acc.c_str(), 0755);
            }
    }
    const std::string tmp = cpath + "." + std::to_string((long)getpid());
    {
        std::ofstream f(tmp, std::ios::binary);
        f.write(code.data(), (std::streamsize)code.size());
    }
    if (rename(tmp.c_str(), cpath.c_str()) != 0) (void)unlink(tmp.c_str());
    return code;
}

JitKernel *get_kernel(int W, int nx, int nu, int N, bool ct, bool fam = false, bool adapt = false) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(cache_mutex());
    const Key key{dev, W, nx, nu, N, (int)ct + 2 * (int)fam + 4 * (int)adapt};
    auto it = cache().find(key);
    if (it != cache().end()) return it->second.failed ? nullptr : &it->second;
    JitKernel k;
    const JitPlan pl = plan_for(W, nx, nu, N, ct, fam, adapt);
    std::string why;
    if (!pl.ok) {
        k.failed = true;
    } else {
        hipDeviceProp_t prop;
        std::string arch = "gfx950";
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0]) {
            arch = prop.gcnArchName;
            const size_t colon = arch.find(':');  // "gfx950:sramecc+:xnack-"
            if (colon != std::string::npos) arch = arch.substr(0, colon);
        }
        // first the disk cache; an image from there that does not load (truncated file, other driver) is compiled again
        for (int attempt = 0; attempt < 2; ++attempt) {
            k.image = build_code_object(pl, nx, nu, N, ct, fam, adapt, arch, attempt == 0, why);
            k.failed = k.image.empty() || hipModuleLoadData(&k.mod, k.image.data()) != hipSuccess ||
                       hipModuleGetFunction(&k.fn, k.mod, "tinympc_jit_solve") != hipSuccess;
            if (!k.failed || k.image.empty()) break;
            (void)hipGetLastError();
            if (k.mod) (void)hipModuleUnload(k.mod);
            k.mod = nullptr;
            k.fn = nullptr;
        }
        k.wpg = pl.wpg;
        if (k.failed && why.empty()) why = "loading the compiled module failed";
    }
    if (k.failed && !why.empty() && getenv("TINYMPC_JIT_VERBOSE"))
        fprintf(stderr, "tinympc-hip: no run-time specialisation for nx=%d nu=%d N=%d: %s\n", nx, nu, N, why.c_str());
    auto ins = cache().emplace(key, std::move(k));
    return ins.first->second.failed ? nullptr : &ins.first->second;
}

}  // namespace

bool solve_jit_supported(int W, int nx, int nu, int N, bool const_tables, bool families, bool adaptive) {
    if (!jit_enabled()) return false;
    if (!plan_for(W, nx, nu, N, const_tables, families, adaptive).ok) return false;
    // compile now, so that a failure is known before the layout is chosen
    return get_kernel(W, nx, nu, N, const_tables, families, adaptive) != nullptr;
}

size_t solve_jit_lds_bytes(int W, int nx, int nu, int N, bool const_tables, bool families, bool adaptive) {
    return plan_for(W, nx, nu, N, const_tables, families, adaptive).lds_bytes;
}

int solve_jit_workgroups(int W, int nx, int nu, int N, bool const_tables, int groups, bool families, bool adaptive) {
    const int wpg = plan_for(W, nx, nu, N, const_tables, families, adaptive).wpg;
    return (groups + wpg - 1) / wpg;
}

hipError_t launch_solve_jit(const SolveParams &p, int W, hipStream_t stream) {
    JitKernel *k = get_kernel(W, p.nx, p.nu, p.N, p.const_tables != 0, p.families != 0, p.adaptive != 0);
    if (!k) return hipErrorInvalidValue;
    SolveParams arg = p;
    void *args[] = {&arg};
    const int wgs = (p.groups + k->wpg - 1) / k->wpg;
    return hipModuleLaunchKernel(k->fn, (unsigned)wgs, 1, 1, 64u * k->wpg, 1, 1, 0, stream, args, nullptr);
}

}  // namespace tinympc
