// C ABI of libdsge_hip.so (declared in include/dsge_hip.h): argument checking, kernel
// dispatch on the tile size BS = ceil(n/8), library-owned scratch, host staging twins.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "dsge_host.hpp"

namespace {
thread_local std::string g_last_error;
}

namespace dsge_host {
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
// Compiled-in defaults (never written after load).  One process-level override, read once when the library is loaded: the
// environment variable DSGE_GENSYS_DOUBLING = 0 | 1 | 2 | 3 replaces the default of dsge_options.gensys_doubling (a site that wants the
// ordered QZ for every draw everywhere without touching its callers; the test suite runs its gensys tests under both values).
static Options defaults_from_environment() {
  Options o{};
  if (const char* e = std::getenv("DSGE_GENSYS_DOUBLING"))
    if ((e[0] >= '0' && e[0] <= '3') && e[1] == '\0') o.gensys_doubling = e[0] - '0';
  // (A/B switch of the round-6 fused assembly + adjoint launch of the gradient pipeline; not part of dsge_options)
  if (const char* e = std::getenv("DSGE_GRAD_FUSED_ADJOINT"))
    if ((e[0] == '0' || e[0] == '1') && e[1] == '\0') o.grad_fused_adjoint = e[0] - '0';
  return o;
}
const Options g_defaults = defaults_from_environment();
thread_local const Options* t_call_options = nullptr;  // options of the call running on this thread
}  // namespace dsge_host

using namespace dsge_host;

namespace {




bool g_device_checked = false;
int g_device_ok = 0;
std::mutex g_mutex;

// Lazy device check (fork-aware: nothing touches HIP before the first call).
int ensure_device() {
  std::lock_guard<std::mutex> lk(g_mutex);
  if (g_device_checked) return g_device_ok ? DSGE_SUCCESS : fail(DSGE_ERR_HIP, "no gfx950 device available");
  g_device_checked = true;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    (void)hipGetLastError();
    return fail(DSGE_ERR_HIP, "no HIP device visible: libdsge_hip has no CPU fallback");
  }
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(DSGE_ERR_HIP, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  g_device_ok = 1;
  return DSGE_SUCCESS;
}

// ---- library-owned device scratch (grown on demand, one arena per device) ----------------
struct Arena {
  void* ptr = nullptr;
  size_t cap = 0;
  int dev = -1;
  bool leased = false;
};
constexpr int MAX_DEV = 16;

// Host-twin staging.  Host threads call the twins concurrently (ctypes releases the GIL: PyMC / nutpie chains in threads,
// two pytensor Ops), so a twin LEASES a staging arena for the duration of its call from a pool that grows to the number of
// concurrent callers; arenas are never shared between two calls in flight and never freed while leased.
std::mutex g_stage_mutex;
std::vector<Arena*> g_stage_pool;

struct StageLease {
  Arena* a = nullptr;
  ~StageLease() {
    if (a) {
      std::lock_guard<std::mutex> lk(g_stage_mutex);
      a->leased = false;
    }
  }
  int reserve(size_t bytes, void** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) return fail(DSGE_ERR_INVALID, "device index out of range");
    {
      std::lock_guard<std::mutex> lk(g_stage_mutex);
      for (Arena* c : g_stage_pool)  // the largest free arena of this device
        if (!c->leased && c->dev == dev && (!a || c->cap > a->cap)) a = c;
      if (!a) {
        a = new Arena();
        a->dev = dev;
        g_stage_pool.push_back(a);
      }
      a->leased = true;
    }
    if (a->cap < bytes) {  // only this call holds the arena: nothing of it is in flight
      if (a->ptr) {
        HIP_TRY(hipFree(a->ptr));
        a->ptr = nullptr;
        a->cap = 0;
      }
      const size_t cap = bytes + bytes / 4 + 4096;
      HIP_TRY(hipMalloc(&a->ptr, cap));
      a->cap = cap;
    }
    *out = a->ptr;
    return DSGE_SUCCESS;
  }
};
#define STAGE_RESERVE(bytes, out) \
  StageLease stage_lease_;        \
  if ((rc = stage_lease_.reserve((bytes), (out)))) return rc

// Installs the options of one call on the calling thread (see dsge_host.hpp); nests.
struct OptionsGuard {
  Options local;
  const Options* prev;
  explicit OptionsGuard(const dsge_options* o) : prev(t_call_options) {
    if (!o) return;
    local.grad_fused_adjoint = g_defaults.grad_fused_adjoint;  // (an internal switch, not in dsge_options: the process-level value)
    local.cr_compact = o->cr_compact;
    local.cr_fused_selection = o->cr_fused_selection;
    local.cr_deflation = o->cr_deflation;
    local.cr_two_waves = o->cr_two_waves;
    local.n_static_hint = o->n_static_hint;
    local.kalman_order = o->kalman_order;
    local.kalman_tiny = o->kalman_tiny;
    local.kalman_block = o->kalman_block;
    local.kalman_mfma = o->kalman_mfma;
    local.pipeline_chunks = o->pipeline_chunks;
    local.gensys_split = o->gensys_split;
    local.kalman_steady_tol = o->kalman_steady_tol;
    local.kalman_nt_products = o->kalman_nt_products;
    local.cr_fused_deflation = o->cr_fused_deflation;
    local.cr_four_waves = o->cr_four_waves;
    local.gensys_real_stage = o->gensys_real_stage;
    local.gensys_pairs = o->gensys_pairs;
    local.gensys_shape_cache = o->gensys_shape_cache;
    local.kalman_narrow = o->kalman_narrow;
    local.gensys_direct_blocks = o->gensys_direct_blocks;
    local.kalman_head_draws = o->kalman_head_draws;
    local.gensys_doubling = o->gensys_doubling;
    local.kalman_grad_split = o->kalman_grad_split;
    local.ll_constant = o->ll_constant;
    local.mask_d = o->mask_d;
    local.joseph = o->joseph;
    local.jitter_F = o->jitter_F;
    local.jitter_P = o->jitter_P;
    t_call_options = &local;
  }
  ~OptionsGuard() { t_call_options = prev; }
};

int check_options(const dsge_options* o) {
  if (!o) return DSGE_SUCCESS;
  if (o->struct_size != sizeof(dsge_options)) return fail(DSGE_ERR_INVALID, "dsge_options.struct_size mismatch (call dsge_options_init)");
  if (!(o->kalman_steady_tol >= 0.0) || o->kalman_steady_tol > 1e-6)
    return fail(DSGE_ERR_INVALID, "kalman_steady_tol must be in [0, 1e-6]");
  if (o->kalman_order < 0 || o->kalman_order > 2) return fail(DSGE_ERR_INVALID, "kalman_order must be 0, 1 or 2");
  if (o->pipeline_chunks < 0 || o->pipeline_chunks > 64) return fail(DSGE_ERR_INVALID, "pipeline_chunks must be in 0..64");
  if (o->gensys_split < 0 || o->gensys_split > 2) return fail(DSGE_ERR_INVALID, "gensys_split must be 0, 1 or 2");
  if (o->gensys_doubling < 0 || o->gensys_doubling > 3) return fail(DSGE_ERR_INVALID, "gensys_doubling must be 0, 1, 2 or 3");
  if (o->kalman_grad_split < 0 || o->kalman_grad_split > 2) return fail(DSGE_ERR_INVALID, "kalman_grad_split must be 0, 1 or 2");
  if (o->n_static_hint < -1 || o->n_static_hint > DSGE_MAX_N) return fail(DSGE_ERR_INVALID, "n_static_hint out of range");
  if (o->ll_constant < DSGE_LL_CONST_P || o->ll_constant > DSGE_LL_CONST_ONE)
    return fail(DSGE_ERR_INVALID, "ll_constant must be DSGE_LL_CONST_P, _OBSERVED or _ONE");
  if (o->jitter_F != o->jitter_F || o->jitter_P != o->jitter_P) return fail(DSGE_ERR_INVALID, "jitter_F / jitter_P must not be NaN");
  return DSGE_SUCCESS;
}

}  // namespace

// ---- StreamArenaPool (dsge_host.hpp) ------------------------------------------------------------------------------------------
namespace dsge_host {
namespace {
std::mutex& pool_registry_mutex() {
  static std::mutex m;
  return m;
}
std::vector<StreamArenaPool*>& pool_registry() {
  static std::vector<StreamArenaPool*> v;
  return v;
}
}  // namespace

StreamArenaPool::StreamArenaPool() {
  std::lock_guard<std::mutex> lk(pool_registry_mutex());
  pool_registry().push_back(this);
}

int StreamArenaPool::reserve(size_t bytes, hipStream_t st, void** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu_);
  Slot* a = nullptr;
  for (auto& slot : slots_)
    if (slot.used && slot.dev == dev && slot.stream == st) a = &slot;
  if (!a)
    for (auto& slot : slots_)
      if (!slot.used && (slot.dev == dev || slot.dev < 0) && (!a || slot.cap > a->cap)) a = &slot;  // the largest free one
  if (!a && slots_.size() < MAX_SLOTS) {
    slots_.emplace_back();
    a = &slots_.back();
  }
  if (!a) {  // every slot is owned by a live stream: take the least recently used one over, once nothing is in flight
    HIP_TRY(hipDeviceSynchronize());
    for (auto& slot : slots_)
      if (slot.dev == dev && (!a || slot.stamp < a->stamp)) a = &slot;
    if (!a) return fail(DSGE_ERR_HIP, "scratch arenas exhausted");
  }
  a->used = true;
  a->dev = dev;
  a->stream = st;
  a->stamp = ++clock_;
  if (a->cap < bytes) {
    if (a->ptr) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(a->ptr));
      a->ptr = nullptr;
      a->cap = 0;
    }
    const size_t cap = bytes + bytes / 4 + 4096;
    HIP_TRY(hipMalloc(&a->ptr, cap));
    a->cap = cap;
  }
  *out = a->ptr;
  return DSGE_SUCCESS;
}

void StreamArenaPool::release(hipStream_t st) {
  std::lock_guard<std::mutex> lk(mu_);
  for (auto& slot : slots_)
    if (slot.used && slot.stream == st) slot.used = false;  // (the memory stays with the slot for its next owner)
}

void stream_arenas_release(hipStream_t st) {
  std::lock_guard<std::mutex> lk(pool_registry_mutex());
  for (StreamArenaPool* p : pool_registry()) p->release(st);
}

namespace {
struct ThreadStreams {
  hipStream_t s[2] = {nullptr, nullptr};
  int dev = -1;
  void drop() {
    for (auto& x : s)
      if (x) {
        (void)hipStreamSynchronize(x);
        stream_arenas_release(x);
        (void)hipStreamDestroy(x);
        x = nullptr;
      }
    (void)hipGetLastError();
  }
  ~ThreadStreams() { drop(); }
};
thread_local ThreadStreams t_streams;
}  // namespace

int twin_streams(hipStream_t* s0, hipStream_t* s1) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (t_streams.dev != dev) {  // (streams belong to a device; recreate after a device switch)
    t_streams.drop();
    for (auto& x : t_streams.s) HIP_TRY(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    t_streams.dev = dev;
  }
  *s0 = t_streams.s[0];
  if (s1) *s1 = t_streams.s[1];
  return DSGE_SUCCESS;
}

}  // namespace dsge_host

namespace {
// The stream the verdict of gensys by spectral division runs on, next to the filter on the caller's stream (pipeline(), round 6):
// one per host thread and device, with its fork / join events; released when the thread exits.
struct VerdictStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  int dev = -1;
  void release() {
    if (!s && !fork && !join) return;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    if (dev >= 0) (void)hipSetDevice(dev);
    if (join) (void)hipEventDestroy(join);
    if (fork) (void)hipEventDestroy(fork);
    if (s) (void)hipStreamDestroy(s);
    s = nullptr;
    fork = join = nullptr;
    if (have_cur) (void)hipSetDevice(cur);
  }
  ~VerdictStream() { release(); }
};
thread_local VerdictStream t_verdict;
int verdict_stream(VerdictStream** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (t_verdict.dev != dev || !t_verdict.s || !t_verdict.fork || !t_verdict.join) {
    t_verdict.release();
    t_verdict.dev = dev;
    HIP_TRY(hipStreamCreateWithFlags(&t_verdict.s, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&t_verdict.fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&t_verdict.join, hipEventDisableTiming));
  }
  *out = &t_verdict;
  return DSGE_SUCCESS;
}

// Scratch of the device entry points (one arena per (device, stream), StreamArenaPool)
StreamArenaPool g_scratch_pool;
StreamArenaPool g_aug_big_pool;  // the gathered model of the augmented route beyond 64 states
int scratch_reserve(hipStream_t st, size_t bytes, void** out) { return g_scratch_pool.reserve(bytes, st, out); }

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base((char*)b) {}
  template <typename T>
  T* take(size_t count) {
    T* p = (T*)(base + off);
    off += align256(count * sizeof(T));
    return p;
  }
};

int check_common(int batch, int n, int n_max) {
  if (batch < 0) return fail(DSGE_ERR_INVALID, "batch < 0");
  if (n < 1 || n > n_max) return fail(DSGE_ERR_INVALID, "n out of range (1.." + std::to_string(n_max) + ")");
  return DSGE_SUCCESS;
}

size_t q_elems(int q_mode, int batch, int k) {
  switch (q_mode) {
    case DSGE_Q_DIAG_SHARED: return (size_t)k;
    case DSGE_Q_DIAG_BATCHED: return (size_t)batch * k;
    case DSGE_Q_FULL_SHARED: return (size_t)k * k;
    case DSGE_Q_FULL_BATCHED: return (size_t)batch * k * k;
    default: return 0;
  }
}

// host staging helpers ---------------------------------------------------------------------
struct Stage {
  Carver carver;
  std::vector<std::pair<void*, std::pair<const void*, size_t>>> ups;
  explicit Stage(void* base) : carver(base) {}
};

}  // namespace

extern "C" {

int dsge_abi_version(void) { return DSGE_ABI_VERSION; }
const char* dsge_last_error(void) { return g_last_error.c_str(); }

int dsge_debug_cr_phases(int enable, long long* cycles_out) {
  int rc = ensure_device();
  if (rc) return rc;
  if (enable && !g_cr_dbg) {
    HIP_TRY(hipMalloc((void**)&g_cr_dbg, 8 * sizeof(long long)));
    HIP_TRY(hipMemset(g_cr_dbg, 0, 8 * sizeof(long long)));
  }
  if (cycles_out && g_cr_dbg) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(cycles_out, g_cr_dbg, 8 * sizeof(long long), hipMemcpyDeviceToHost));
  }
  if (!enable && g_cr_dbg) {
    (void)hipFree(g_cr_dbg);
    g_cr_dbg = nullptr;
  }
  return DSGE_SUCCESS;
}
int dsge_debug_second_order_phases(int enable, long long* cycles_out) {
  int rc = ensure_device();
  if (rc) return rc;
  if (enable && !g_so_dbg) {
    HIP_TRY(hipMalloc((void**)&g_so_dbg, 8 * sizeof(long long)));
    HIP_TRY(hipMemset(g_so_dbg, 0, 8 * sizeof(long long)));
  }
  if (cycles_out && g_so_dbg) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(cycles_out, g_so_dbg, 8 * sizeof(long long), hipMemcpyDeviceToHost));
  }
  if (!enable && g_so_dbg) {
    (void)hipFree(g_so_dbg);
    g_so_dbg = nullptr;
  }
  return DSGE_SUCCESS;
}
int dsge_debug_adjoint_refine(int mode) {
  if (mode < 0 || mode > 2) return fail(DSGE_ERR_INVALID, "mode out of range (0..2)");
  g_adj_refine_mode = mode;
  return DSGE_SUCCESS;
}
int dsge_forget_measured_shapes(void) {
  cr_deflation_reset();
  gensys_shape_reset();
  return DSGE_SUCCESS;
}
int dsge_debug_kalman_steady_steps(int32_t* steady_at_device) {
  g_kalman_steady_at = steady_at_device;
  return DSGE_SUCCESS;
}
int dsge_debug_kalman_timeline(long long* timeline_device) {
  g_kalman_timeline = timeline_device;
  return DSGE_SUCCESS;
}

int dsge_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int ok = 0;
  for (int i = 0; i < count; ++i) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

int dsge_set_device(int device) {
  HIP_TRY(hipSetDevice(device));
  return DSGE_SUCCESS;
}

int dsge_stream_synchronize(void* stream) {
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return DSGE_SUCCESS;
}

static int cr_entry(const double* A, const double* B, const double* C, int batch, int n, int max_iter, double tol,
                    double* T_out, int32_t* status, int32_t* n_iter, void* stream, int scan_mode) {
  int rc = check_common(batch, n, DSGE_MAX_N_BIG);
  if (rc) return rc;
  if (!A || !B || !C || !T_out || !status) return fail(DSGE_ERR_INVALID, "null pointer");
  if (max_iter < 0) return fail(DSGE_ERR_INVALID, "max_iter < 0");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  if (big_size(n))  // 65 .. 96 variables: one workgroup per draw (dsge_big.hpp)
    return launch_cr_big(A, B, C, batch, n, max_iter, tol, T_out, status, n_iter, (hipStream_t)stream, scan_mode, nullptr, 0,
                         nullptr);
  return launch_cr(A, B, C, batch, n, max_iter, tol, T_out, status, n_iter, (hipStream_t)stream, scan_mode);
}

int dsge_cycle_reduction_batched(const double* A, const double* B, const double* C, int batch, int n, int max_iter,
                                 double tol, double* T_out, int32_t* status, int32_t* n_iter, void* stream) {
  return cr_entry(A, B, C, batch, n, max_iter, tol, T_out, status, n_iter, stream, 0);
}

int dsge_scan_cycle_reduction_batched(const double* A, const double* B, const double* C, int batch, int n,
                                      int max_iter, double tol, double* T_out, int32_t* status, int32_t* n_steps,
                                      void* stream) {
  return cr_entry(A, B, C, batch, n, max_iter, tol, T_out, status, n_steps, stream, 1);
}

int dsge_gensys_batched(const double* A, const double* B, const double* C, const double* D, int batch, int n, int k,
                        double tol, int n_lead_hint, double* T_out, double* R_out, int32_t* eu_out, int32_t* status,
                        void* stream) {
  // 65 .. 96 variables: by spectral division only (csrc/dsge_big.hpp: gensys_certify_big_kernel; no ordered QZ at that size)
  const bool big = big_size(n) && opt().gensys_doubling != 0;
  int rc = check_common(batch, n, big ? DSGE_MAX_N_BIG : DSGE_MAX_N_GENSYS - 1);
  if (rc) return rc;
  if (!A || !B || !C || !T_out || !eu_out || !status) return fail(DSGE_ERR_INVALID, "null pointer");
  if (R_out && (!D || k < 1 || k > n)) return fail(DSGE_ERR_INVALID, "R_out requires D and 1 <= k <= n");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  if (big) return launch_gensys_big(A, B, C, D, batch, n, k, tol, T_out, R_out, eu_out, status, nullptr, (hipStream_t)stream);
  if ((rc = launch_gensys(A, B, C, batch, n, tol, n_lead_hint, T_out, eu_out, status, (hipStream_t)stream, nullptr, nullptr, nullptr,
                          R_out ? D : nullptr, k, R_out)))
    return rc;
  if (R_out)  // gensys_pt: R = -(C T + B)^-1 D  (gensys.py:681); computed for every draw, as the graph does
    return launch_assemble(nullptr, B, C, D, T_out, nullptr, nullptr, 0, batch, n, k, R_out, nullptr, nullptr, nullptr,
                           nullptr, 1, 0, (hipStream_t)stream);
  return DSGE_SUCCESS;
}

int dsge_gensys_pencil_batched(const double* g0, const double* g1, const double* c, const double* psi, const double* pi,
                               int batch, int N, int k, int n_eta, double tol, double* G1_out, double* C_out,
                               double* impact_out, double* gev_out, int32_t* eu_out, int32_t* status, void* stream) {
  return dsge_gensys_pencil_full_batched(g0, g1, c, psi, pi, batch, N, k, n_eta, tol, G1_out, C_out, impact_out, gev_out,
                                         eu_out, status, nullptr, stream);
}

int dsge_gensys_pencil_full_batched(const double* g0, const double* g1, const double* c, const double* psi,
                                    const double* pi, int batch, int N, int k, int n_eta, double tol, double* G1_out,
                                    double* C_out, double* impact_out, double* gev_out, int32_t* eu_out, int32_t* status,
                                    const dsge_gensys_forward* forward, void* stream) {
  int rc = check_common(batch, N, DSGE_MAX_N_GENSYS);
  if (rc) return rc;
  if (k < 1 || n_eta < 0 || n_eta + k + 1 > 64) return fail(DSGE_ERR_INVALID, "need k >= 1, n_eta >= 0, n_eta + k + 1 <= 64");
  if (!g0 || !g1 || !psi || (n_eta > 0 && !pi) || !G1_out || !C_out || !impact_out || !gev_out || !eu_out || !status)
    return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  if (forward && forward->loose && n_eta < 1) return fail(DSGE_ERR_INVALID, "loose needs n_eta >= 1");
  return launch_gensys_pencil(g0, g1, c, psi, pi, batch, N, k, n_eta, tol, G1_out, C_out, impact_out, gev_out, eu_out, status,
                              (hipStream_t)stream, forward);
}

int dsge_bk_eigenvalues_batched(const double* A, const double* B, const double* C, int batch, int n, double tol,
                                double* eig_re, double* eig_im, int32_t* n_eig, int32_t* n_forward,
                                int32_t* n_unstable, int32_t* status, void* stream) {
  int rc = check_common(batch, n, DSGE_MAX_N_GENSYS - 1);
  if (rc) return rc;
  if (!A || !B || !C || !eig_re || !eig_im || !n_eig || !n_forward || !n_unstable || !status)
    return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  return launch_gensys_bk(A, B, C, batch, n, tol, eig_re, eig_im, n_eig, n_forward, n_unstable, status,
                          (hipStream_t)stream);
}

int dsge_selection_batched(const double* A, const double* B, const double* C, const double* D, const double* T,
                           int batch, int n, int k, double* R_out, double* resid_out, void* stream) {
  int rc = check_common(batch, n, DSGE_MAX_N_BIG);
  if (rc) return rc;
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "k out of range (1..n)");
  if (!B || !C || !D || !T || !R_out) return fail(DSGE_ERR_INVALID, "null pointer");
  if (resid_out && !A) return fail(DSGE_ERR_INVALID, "resid_out requires A");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  if (big_size(n)) return launch_selection_big(A, B, C, D, T, batch, n, k, R_out, resid_out, nullptr, (hipStream_t)stream);
  return launch_assemble(A, B, C, D, T, nullptr, nullptr, 0, batch, n, k, R_out, resid_out, nullptr, nullptr, nullptr,
                         1, 0, (hipStream_t)stream);
}

int dsge_policy_adjoints_batched(const double* B, const double* C, const double* T, const double* T_bar, int batch,
                                 int n, double* A_bar, double* B_bar, double* C_bar, int32_t* status, void* stream) {
  int rc = check_common(batch, n, 56);  // [M' | T_bar | C'] + two operands in LDS: n <= 56 (132 KB at the 64-wide tile, not built)
  if (rc) return rc;
  if (!B || !C || !T || !T_bar || !A_bar || !B_bar || !C_bar || !status) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  return launch_adjoint(B, C, T, T_bar, batch, n, A_bar, B_bar, C_bar, status, (hipStream_t)stream);
}

int dsge_selection_adjoints_batched(const double* B, const double* C, const double* T, const double* R, const double* R_bar,
                                    int batch, int n, int k, double* B_bar, double* C_bar, double* D_bar, double* T_bar,
                                    void* stream) {
  int rc = check_common(batch, n, 56);  // (grad_assemble_kernel: 153 KB of LDS at the 56-wide tile, 199 KB at 64)
  if (rc) return rc;
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "k out of range (1..n)");
  if (!B || !C || !T || !R || !R_bar || !B_bar || !C_bar || !D_bar || !T_bar) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  return launch_grad_assemble(B, C, T, R, nullptr, 0, nullptr, batch, n, k, nullptr, T_bar, B_bar, C_bar, D_bar, nullptr,
                              (hipStream_t)stream, R_bar);
}

int dsge_policy_norms_batched(const double* A, const double* B, const double* C, const double* D, const double* T,
                              const double* R, const int32_t* state_mask, int batch, int n, int k,
                              double* det_norm_out, double* stoch_norm_out, void* stream) {
  int rc = check_common(batch, n, 56);  // five n x n matrices in LDS: n <= 56
  if (rc) return rc;
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "k out of range (1..n)");
  if (!A || !B || !C || !D || !T || !R || !state_mask || !det_norm_out || !stoch_norm_out)
    return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  return launch_norms(A, B, C, D, T, R, state_mask, batch, n, k, det_norm_out, stoch_norm_out, (hipStream_t)stream);
}

int dsge_backward_direct_batched(const double* A, const double* B, const double* D, int batch, int n, int k,
                                 double* T_out, double* R_out, void* stream) {
  int rc = check_common(batch, n, DSGE_MAX_N_CR);
  if (rc) return rc;
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "k out of range (1..n)");
  if (!A || !B || !D || !T_out || !R_out) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  return launch_bdirect(A, B, D, batch, n, k, T_out, R_out, (hipStream_t)stream);
}

int dsge_lyapunov_batched(const double* T, const double* R, const double* Q, int q_mode, int batch, int m, int k,
                          double* P0_out, double* RQR_out, int32_t* status, void* stream) {
  int rc = check_common(batch, m, DSGE_MAX_N);
  if (rc) return rc;
  if (k < 1 || k > m) return fail(DSGE_ERR_INVALID, "k out of range (1..m)");
  if (q_mode < 0 || q_mode > 3) return fail(DSGE_ERR_INVALID, "bad q_mode");
  if (!T || !R || !Q || !P0_out || !status) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  HIP_TRY(hipMemsetAsync(status, 0, sizeof(int32_t) * batch, (hipStream_t)stream));
  return launch_assemble(nullptr, nullptr, nullptr, nullptr, T, R, Q, q_mode, batch, m, k, nullptr, nullptr, RQR_out,
                         P0_out, status, 0, 1, (hipStream_t)stream);
}

int dsge_autocorrelation_batched(const double* T, const double* R, const double* Q, int q_mode, const double* Z,
                                 const double* Hdiag, int batch, int m, int k, int p, int n_lags, int lag_step,
                                 int correlation, double* acf_out, double* Sigma_out, int32_t* status, void* stream) {
  int rc = check_common(batch, m, DSGE_MAX_N);
  if (rc) return rc;
  if (k < 1 || k > m) return fail(DSGE_ERR_INVALID, "k out of range (1..m)");
  if (q_mode < 0 || q_mode > 3) return fail(DSGE_ERR_INVALID, "bad q_mode");
  if (n_lags < 0 || lag_step < 1) return fail(DSGE_ERR_INVALID, "n_lags >= 0 and lag_step >= 1 required");
  if (Z && (p < 1 || p > DSGE_MAX_P)) return fail(DSGE_ERR_INVALID, "p out of range (1..DSGE_MAX_P)");
  if (!T || !R || !Q || !acf_out || !status) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  hipStream_t st = (hipStream_t)stream;
  double* Sigma = Sigma_out;
  if (!Sigma) {
    void* base = nullptr;
    if ((rc = scratch_reserve(st, align256((size_t)batch * m * m * sizeof(double)) + 4096, &base))) return rc;
    Sigma = (double*)base;
  }
  HIP_TRY(hipMemsetAsync(status, 0, sizeof(int32_t) * batch, st));
  if ((rc = launch_assemble(nullptr, nullptr, nullptr, nullptr, T, R, Q, q_mode, batch, m, k, nullptr, nullptr, nullptr,
                            Sigma, status, 0, 1, st)))
    return rc;
  return launch_acf(T, Sigma, Z, Hdiag, batch, m, p, n_lags, lag_step, correlation, acf_out, status, st);
}

int dsge_kalman_filter_outputs_batched(const double* T, const double* R, const double* Q, int q_mode, const double* Z,
                                       int z_batched, const double* d, int d_batched, const double* Hdiag, int h_batched,
                                       const double* y, int batch, int m, int k, int p, int T_len, double jitter,
                                       double missing_fill, double* ll_out, double* a_pred_out, double* a_filt_out,
                                       double* p_pred_out, double* p_filt_out, int full_cov, int32_t* status_io, void* stream) {
  int rc = check_common(batch, m, DSGE_MAX_N);
  if (rc) return rc;
  if (k < 1 || k > m) return fail(DSGE_ERR_INVALID, "k out of range (1..m)");
  if (p < 1 || p > DSGE_MAX_P) return fail(DSGE_ERR_INVALID, "p out of range (1..DSGE_MAX_P)");
  if (T_len < 0) return fail(DSGE_ERR_INVALID, "T_len < 0");
  if (q_mode < 0 || q_mode > 3) return fail(DSGE_ERR_INVALID, "bad q_mode");
  if (!T || !R || !Q || !Z || !y || !ll_out || !status_io) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  hipStream_t st = (hipStream_t)stream;
  const size_t mm = (size_t)batch * m * m;
  void* base = nullptr;
  if ((rc = scratch_reserve(st, 2 * align256(mm * 8) + 4096, &base))) return rc;
  Carver cv(base);
  double* RQR = cv.take<double>(mm);
  double* P0 = cv.take<double>(mm);
  if ((rc = launch_assemble(nullptr, nullptr, nullptr, nullptr, T, R, Q, q_mode, batch, m, k, nullptr, nullptr, RQR, P0,
                            status_io, 0, 1, st)))
    return rc;
  return launch_kalman_outputs(T, RQR, P0, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, m, p, T_len, jitter,
                               missing_fill, ll_out, a_pred_out, a_filt_out, p_pred_out, p_filt_out, full_cov, status_io, st);
}

int dsge_kalman_logp_batched(const double* T, const double* R, const double* Q, int q_mode, const double* Z,
                             int z_batched, const double* d, int d_batched, const double* Hdiag, int h_batched,
                             const double* y, int batch, int m, int k, int p, int T_len, double jitter,
                             double missing_fill, int n_state_hint, int z_selector_hint, double* logp_out,
                             int32_t* status_io, void* stream) {
  int rc = check_common(batch, m, DSGE_MAX_N);
  if (rc) return rc;
  if (k < 1 || k > m) return fail(DSGE_ERR_INVALID, "k out of range (1..m)");
  if (p < 1 || p > DSGE_MAX_P) return fail(DSGE_ERR_INVALID, "p out of range (1..DSGE_MAX_P)");
  if (T_len < 0) return fail(DSGE_ERR_INVALID, "T_len < 0");
  if (q_mode < 0 || q_mode > 3) return fail(DSGE_ERR_INVALID, "bad q_mode");
  if (!T || !R || !Q || !Z || !y || !logp_out || !status_io) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  hipStream_t st = (hipStream_t)stream;
  void* base = nullptr;
  const size_t mm = (size_t)batch * m * m;
  if ((rc = scratch_reserve(st, 2 * align256(mm * sizeof(double)) + 4096, &base))) return rc;
  Carver cv(base);
  double* RQR = cv.take<double>(mm);
  double* P0 = cv.take<double>(mm);
  if ((rc = launch_assemble(nullptr, nullptr, nullptr, nullptr, T, R, Q, q_mode, batch, m, k, nullptr, nullptr, RQR, P0,
                            status_io, 0, 2, st)))
    return rc;
  return launch_kalman(T, RQR, P0, 0, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, m, p, T_len, jitter,
                       missing_fill, n_state_hint, z_selector_hint, logp_out, status_io, st);
}

inline size_t pipeline_scratch_bytes(int batch, int n, int k) {
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k;
  size_t b = 3 * align256(nn * 8) + align256(nk * 8) + align256((size_t)batch * 12) + 3 * align256((size_t)batch * 4) +
             align256((size_t)batch * 8) + 4096;
  if (n > 64)  // pipeline_big: the gathered model (T_r, R_r, Z_r at 64 filtered variables) next to the full-size T, R
    b += align256((size_t)batch * 64 * 64 * 8) + align256((size_t)batch * 64 * k * 8) +
         align256((size_t)batch * DSGE_MAX_P * 64 * 8) + 1024;
  return b;
}

// ---- 65 .. 96 variables (dsge_big.hpp): cycle reduction with one workgroup per draw, then the EXISTING filter kernels on the
// model restricted to F = {state variables} u {observed variables}: x_t[F] = T[F, F] x_{t-1}[F] + R[F, :] eps_t is exact because
// every column of T outside the state variables is zero (T = -(B + C T)^-1 A inherits the zero columns of A), and y_t reads
// x_t[F] only.  |F| <= 64 or DSGE_ERR_TOO_LARGE.  Measuring F synchronises the stream once per call.
static int pipeline_big(const double* A, const double* B, const double* C, const double* D, const double* Q, int q_mode,
                        const double* Z, int z_batched, const double* d, int d_batched, const double* Hdiag, int h_batched,
                        const double* y, int batch, int n, int k, int p, int T_len, int solver, bool park_failures, double tol,
                        int max_iter, double jitter, double missing_fill, int z_selector_hint, double* logp_out,
                        int32_t* status_out, double* T_out, double* R_out, double* resid_out, int32_t* n_iter_out, hipStream_t st,
                        int reps, float* ms_out, void* scratch_slice) {
  int rc;
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k;
  void* base = scratch_slice;
  if (!base && (rc = scratch_reserve(st, pipeline_scratch_bytes(batch, n, k), &base))) return rc;
  Carver cv(base);
  double* Tw = T_out ? T_out : cv.take<double>(nn);
  double* Rw = R_out ? R_out : cv.take<double>(nk);
  double* T_r = cv.take<double>((size_t)batch * 64 * 64);
  double* R_r = cv.take<double>((size_t)batch * 64 * k);
  double* Z_r = cv.take<double>(z_batched ? (size_t)batch * p * 64 : (size_t)p * 64);
  double* RQR = cv.take<double>((size_t)batch * 64 * 64);
  double* P0 = cv.take<double>((size_t)batch * 64 * 64);
  int32_t* it_w = n_iter_out ? n_iter_out : cv.take<int32_t>((size_t)batch);
  int32_t* park_w = cv.take<int32_t>((size_t)batch);
  int32_t* eu_w = cv.take<int32_t>((size_t)batch * 3);  // (gensys: the certificate's eu codes)
  // F depends on A and Z only: measured FIRST (one small launch, a 72-byte read-back, one stream synchronisation per call -- not
  // per repetition), so that a model with more than 64 state + observed variables is refused before anything is enqueued and
  // before any output of this call is touched (DSGE_ERR_TOO_LARGE: "nothing computed", include/dsge_hip.h)
  unsigned char idx[64];
  int u = 0, ns = 0;
  if ((rc = big_filtered_variables(A, Z, z_batched, batch, n, p, st, idx, &u, &ns))) return rc;
  if (u > 64)
    return fail(DSGE_ERR_TOO_LARGE, "solve + Kalman with n > 64: " + std::to_string(u) +
                                        " state and observed variables, the filter kernels take at most 64");
  if (u < 1) return fail(DSGE_ERR_INVALID, "solve + Kalman with n > 64: no state and no observed variable");
  EventGuard ev[4];
  float acc_ms[3] = {0.f, 0.f, 0.f};
  if (ms_out)
    for (auto& e : ev) HIP_TRY(e.create());
  const int n_rep = ms_out ? reps : 1;
  for (int rep = 0; rep < n_rep; ++rep) {
    if (ms_out) HIP_TRY(hipEventRecord(ev[0], st));
    const bool scan = solver == DSGE_SOLVER_SCAN_CYCLE_REDUCTION;
    // R from the final elimination of the (njit-rule) iteration; the scan variant and a requested residual take the explicit
    // selection R = -(C T + B)^-1 D
    // (nor with the reference's default gating: a failed draw carries T = 0 on, and its R = -(C 0 + B)^-1 D comes from the
    // explicit selection, which runs after the statuses have been parked)
    const bool fuse_R = !scan && !resid_out && !park_failures;
    if (solver == DSGE_SOLVER_GENSYS) {
      // gensys at this size = the doubling iteration + the certificate of eu = [1, 1, 0] (gensys_certify_big_kernel), R from the
      // certificate's elimination; a draw without the certificate is a failed draw (no ordered QZ beyond 64 variables)
      if ((rc = launch_gensys_big(A, B, C, D, batch, n, k, tol, Tw, Rw, eu_w, status_out, it_w, st))) return rc;
      if (resid_out && (rc = launch_selection_big(A, B, C, D, Tw, batch, n, k, nullptr, resid_out, status_out, st))) return rc;
    } else {
    if ((rc = launch_cr_big(A, B, C, batch, n, max_iter, tol, Tw, status_out, it_w, st, scan ? 1 : 0, fuse_R ? D : nullptr, k,
                            fuse_R ? Rw : nullptr)))
      return rc;
    if (park_failures && (rc = launch_status_park(status_out, park_w, batch, 0, st))) return rc;
    if (!fuse_R && (rc = launch_selection_big(A, B, C, D, Tw, batch, n, k, Rw, resid_out, status_out, st))) return rc;
    }
    if (ms_out) HIP_TRY(hipEventRecord(ev[1], st));
    if ((rc = launch_big_compress(Tw, Rw, Z, z_batched, batch, n, k, p, idx, u, T_r, R_r, Z_r, st))) return rc;
    const int ns_hint = (ns > 0 && ns < u) ? ns : 0;
    const bool q_diag = q_mode == DSGE_Q_DIAG_SHARED || q_mode == DSGE_Q_DIAG_BATCHED;
    const bool fold_rqr = !park_failures && q_diag && kalman_folds_rqr(u, p, k, ns_hint, z_selector_hint);
    if (fold_rqr)
      rc = DSGE_SUCCESS;
    else if (q_diag && k <= 16)
      rc = launch_rqr(R_r, Q, q_mode == DSGE_Q_DIAG_BATCHED, batch, u, k, status_out, RQR, st);
    else
      rc = launch_assemble(nullptr, nullptr, nullptr, nullptr, T_r, R_r, Q, q_mode, batch, u, k, nullptr, nullptr, RQR, P0,
                           status_out, 0, 2, st);
    if (rc) return rc;
    if (ms_out) HIP_TRY(hipEventRecord(ev[2], st));
    const int32_t* okey = (opt().kalman_order == 1) ? it_w : nullptr;
    if ((rc = launch_kalman(T_r, RQR, P0, 0, Z_r, z_batched, d, d_batched, Hdiag, h_batched, y, batch, u, p, T_len, jitter,
                            missing_fill, ns_hint, z_selector_hint, logp_out, status_out, st, okey, fold_rqr ? R_r : nullptr,
                            fold_rqr ? Q : nullptr, q_mode == DSGE_Q_DIAG_BATCHED, k, nullptr)))
      return rc;
    if (park_failures && (rc = launch_status_park(status_out, park_w, batch, 1, st))) return rc;
    if (ms_out) {
      HIP_TRY(hipEventRecord(ev[3], st));
      HIP_TRY(hipEventSynchronize(ev[3]));
      for (int i = 0; i < 3; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        acc_ms[i] += ms;
      }
    }
  }
  if (ms_out)
    for (int i = 0; i < 3; ++i) ms_out[i] = acc_ms[i] / (float)n_rep;
  return DSGE_SUCCESS;
}

static int pipeline(const double* A, const double* B, const double* C, const double* D, const double* Q, int q_mode,
                    const double* Z, int z_batched, const double* d, int d_batched, const double* Hdiag, int h_batched,
                    const double* y, int batch, int n, int k, int p, int T_len, int solver, double tol, int max_iter,
                    double jitter, double missing_fill, int n_state_hint, int z_selector_hint, int n_lead_hint,
                    double* logp_out, int32_t* status_out, double* T_out, double* R_out, double* resid_out,
                    int32_t* n_iter_out, hipStream_t st, int reps, float* ms_out, int arena_id = 0,
                    void* scratch_slice = nullptr) {
  (void)arena_id;  // (scratch is keyed by stream now; the chunked host path runs its chunks on two streams)
  const bool zero_T_on_failure = (solver & DSGE_SOLVER_FLAG_ZERO_T_ON_FAILURE) != 0;
  solver &= ~DSGE_SOLVER_FLAG_ZERO_T_ON_FAILURE;
  const bool is_cr = solver == DSGE_SOLVER_CYCLE_REDUCTION || solver == DSGE_SOLVER_SCAN_CYCLE_REDUCTION;
  const bool park_failures = zero_T_on_failure && is_cr;
  // (gensys beyond 64 variables exists by spectral division only: dsge_options.gensys_doubling != 0)
  const bool big_ok = is_cr || (solver == DSGE_SOLVER_GENSYS && opt().gensys_doubling != 0);
  int rc = check_common(batch, n, big_ok ? DSGE_MAX_N_BIG : DSGE_MAX_N);
  if (rc) return rc;
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "k out of range (1..n)");
  if (p < 1 || p > DSGE_MAX_P) return fail(DSGE_ERR_INVALID, "p out of range (1..DSGE_MAX_P)");
  if (T_len < 0) return fail(DSGE_ERR_INVALID, "T_len < 0");
  if (q_mode < 0 || q_mode > 3) return fail(DSGE_ERR_INVALID, "bad q_mode");
  if (!A || !B || !C || !D || !Q || !Z || !y || !logp_out || !status_out) return fail(DSGE_ERR_INVALID, "null pointer");
  if (solver != DSGE_SOLVER_CYCLE_REDUCTION && solver != DSGE_SOLVER_BACKWARD_DIRECT && solver != DSGE_SOLVER_GENSYS &&
      solver != DSGE_SOLVER_SCAN_CYCLE_REDUCTION)
    return fail(DSGE_ERR_INVALID, "unknown solver code");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  if (big_size(n))
    return pipeline_big(A, B, C, D, Q, q_mode, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, n, k, p, T_len, solver,
                        park_failures, tol, max_iter, jitter, missing_fill, z_selector_hint, logp_out, status_out, T_out, R_out,
                        resid_out, n_iter_out, st, reps, ms_out, scratch_slice);

  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k;
  void* base = scratch_slice;  // (a slice of an arena the caller reserved: chunks in flight on several streams)
  if (!base &&
      (rc = scratch_reserve(st, pipeline_scratch_bytes(batch, n, k), &base)))
    return rc;
  Carver cv(base);
  double* Tw = T_out ? T_out : cv.take<double>(nn);
  double* Rw = R_out ? R_out : cv.take<double>(nk);
  double* RQR = cv.take<double>(nn);
  double* P0 = cv.take<double>(nn);
  int32_t* eu_w = cv.take<int32_t>((size_t)batch * 3);
  int32_t* it_w = n_iter_out ? n_iter_out : cv.take<int32_t>((size_t)batch);  // cycle-reduction iterations
  int32_t* key_w = cv.take<int32_t>((size_t)batch);                            // dispatch key of the Kalman launch
  unsigned long long* cm_w = cv.take<unsigned long long>((size_t)batch);       // non-zero columns of T, from the solver
  int32_t* vst_w = cv.take<int32_t>((size_t)batch);                            // status words of an overlapped gensys verdict

  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  float acc_ms[3] = {0.f, 0.f, 0.f};
  if (ms_out)
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
  const int n_rep = ms_out ? reps : 1;
  for (int rep = 0; rep < n_rep; ++rep) {
    if (ms_out) HIP_TRY(hipEventRecord(ev[0], st));
    // Cycle reduction (njit semantics) can hand back R from its final elimination (A1_hat = B + C T at convergence);
    // the explicit R = -(C T + B)^-1 D of the assemble kernel is kept whenever the caller wants the policy residual.
    const bool fuse_R = (solver == DSGE_SOLVER_CYCLE_REDUCTION) && !resid_out && opt().cr_fused_selection && !park_failures;
    bool have_colmask = false;
    int gensys_key = 0;  // 1: the gensys launches have written the Kalman dispatch key
    const int32_t* gensys_qz_marks = nullptr;  // gensys by spectral division: Rw already holds R of the unmarked draws
    dsge_host::GensysOverlap gov;              // ... with the verdict on a second stream (used = 1: join, merge, second filter pass below)
    hipEvent_t gov_join = nullptr;
    if (is_cr) {
      int deflated = 0;
      // static variables deflated first (the iteration then runs on n - h variables); not when the caller asks for the
      // iteration counts or the residual, whose contract is the full-size iteration
      if (fuse_R && !n_iter_out &&
          (rc = launch_cr_deflated(A, B, C, D, batch, n, k, max_iter, tol, Tw, Rw, status_out, it_w, st, &deflated, cm_w)))
        return rc;
      have_colmask = deflated == 2;
      if (!deflated)
        rc = launch_cr(A, B, C, batch, n, max_iter, tol, Tw, status_out, it_w, st,
                       solver == DSGE_SOLVER_SCAN_CYCLE_REDUCTION ? 1 : 0, fuse_R ? D : nullptr, k, fuse_R ? Rw : nullptr);
    } else if (solver == DSGE_SOLVER_GENSYS) {
      if (n_iter_out) HIP_TRY(hipMemsetAsync(n_iter_out, 0, sizeof(int32_t) * batch, st));
      // Round 6: gensys by spectral division with the VERDICT -- certificate, compaction, the ordered QZ of the draws without one:
      // 0.38 ms + six (mostly empty) launches per 4096 draws -- on a second stream NEXT to the filter: the filter of every draw
      // starts behind the doubling iteration with that iteration's T and R; afterwards the draws the verdict re-solved are filtered
      // again (second passes) and the ones it rejected get their status and logp = -inf.  A certified draw -- every draw of an
      // estimation run -- is filtered once, with exactly the inputs of the serial order.  Only in the plain fused evaluation
      // (no residual, no profiling, R folded into the filter, not while the caller captures the stream); dsge_options.
      // gensys_doubling = 3 keeps the verdict on the caller's stream.
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      const bool capturing = (hipStreamIsCapturing(st, &cap) == hipSuccess) && cap != hipStreamCaptureStatusNone;
      const bool q_diag_ = q_mode == DSGE_Q_DIAG_SHARED || q_mode == DSGE_Q_DIAG_BATCHED;
      const bool want_overlap = opt().gensys_doubling == 1 && !ms_out && !resid_out && !park_failures && !capturing && batch >= 256 &&
                                q_diag_ && n <= 64 && kalman_folds_rqr(n, p, k, n_state_hint, z_selector_hint);
      VerdictStream* vs = nullptr;
      if (want_overlap) {
        if ((rc = verdict_stream(&vs))) return rc;
        gov.st = vs->s;
        gov.fork = vs->fork;
        gov.status = vst_w;
      }
      rc = launch_gensys(A, B, C, batch, n, tol, n_lead_hint, Tw, eu_w, status_out, st, nullptr,
                         (opt().kalman_order != 0 && batch >= 512) ? key_w : nullptr, &gensys_key, D, k, Rw, n_state_hint,
                         &gensys_qz_marks, want_overlap ? &gov : nullptr);
      if (!rc && gov.used) gov_join = vs->join;
    } else {
      HIP_TRY(hipMemsetAsync(status_out, 0, sizeof(int32_t) * batch, st));
      if (n_iter_out) HIP_TRY(hipMemsetAsync(n_iter_out, 0, sizeof(int32_t) * batch, st));
      rc = launch_bdirect(A, B, D, batch, n, k, Tw, Rw, st);
    }
    if (rc) return rc;
    if (park_failures) {  // failed draws carry T = 0 on as ordinary draws (the column-mask buffer is free: no deflated solve in this mode)
      if ((rc = launch_status_park(status_out, (int32_t*)cm_w, batch, 0, st))) return rc;
    }
    if (ms_out) HIP_TRY(hipEventRecord(ev[1], st));
    // backward_direct already produced R; the assemble kernel recomputes it from the same
    // formula (C T + B = B when C == 0), which keeps a single code path for resid/RQR/P0.
    const bool q_diag = q_mode == DSGE_Q_DIAG_SHARED || q_mode == DSGE_Q_DIAG_BATCHED;
    // sym(R Q R') inside the filter kernel (its retained block only) when the fast selector kernel takes the draws
    // (round 4: also behind the explicit selection of gensys / backward-direct / a requested residual -- the assemble launch then
    // forms R (and the residual) only: no 40 x 40 product, no 51 MB of R Q R' through HBM per 4096 draws)
    const bool fold_rqr = !park_failures && q_diag && n <= 64 && kalman_folds_rqr(n, p, k, n_state_hint, z_selector_hint);
    if (fold_rqr && fuse_R)
      rc = DSGE_SUCCESS;
    else if (fold_rqr && gov.used)
      rc = DSGE_SUCCESS;  // (R of the draws the verdict re-solves: behind the join, below)
    else if (fold_rqr)
      rc = launch_assemble(A, B, C, D, Tw, nullptr, Q, q_mode, batch, n, k, Rw, resid_out, nullptr, nullptr, status_out, 1, 0, st,
                           resid_out ? nullptr : gensys_qz_marks);
    else if (fuse_R && q_diag && k <= 16 && n <= 64)
      rc = launch_rqr(Rw, Q, q_mode == DSGE_Q_DIAG_BATCHED, batch, n, k, status_out, RQR, st);  // (RQR_KMAX = 16)
    else if (fuse_R)
      rc = launch_assemble(nullptr, nullptr, nullptr, nullptr, Tw, Rw, Q, q_mode, batch, n, k, nullptr, nullptr, RQR, P0,
                           status_out, 0, 2, st);
    else
      rc = launch_assemble(A, B, C, D, Tw, nullptr, Q, q_mode, batch, n, k, Rw, resid_out, RQR, P0, status_out, 1, 2, st);
    if (rc) return rc;
    if (ms_out) HIP_TRY(hipEventRecord(ev[2], st));
    const int32_t* okey = nullptr;
    // dispatch key of the Kalman launch: the cycle-reduction iteration counts when there are any (free, and the better
    // predictor on the bench workload), else a persistence key computed from T itself (gensys, backward-direct)
    if (opt().kalman_order == 1 && is_cr) {
      okey = it_w;
    } else if (opt().kalman_order != 0 && batch >= 512 && n <= 64) {
      // gensys on the window path has written the key from its own spectrum (gensys_post_kernel)
      if (!gensys_key && (rc = launch_persistence_key(Tw, status_out, batch, n, key_w, st))) return rc;
      okey = key_w;
    }
    if ((rc = launch_kalman(Tw, RQR, P0, 0, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, n, p, T_len, jitter,
                            missing_fill, n_state_hint, z_selector_hint, logp_out, status_out, st, okey,
                            fold_rqr ? Rw : nullptr, fold_rqr ? Q : nullptr, q_mode == DSGE_Q_DIAG_BATCHED, k,
                            have_colmask ? cm_w : nullptr)))
      return rc;
    if (gov.used) {
      // join the verdict; R of the draws the ordered QZ solved (explicit selection, marked draws only, by the verdict's status);
      // their status words; their filter (every launch a second pass)
      HIP_TRY(hipEventRecord(gov_join, gov.st));
      HIP_TRY(hipStreamWaitEvent(st, gov_join, 0));
      if ((rc = launch_assemble(A, B, C, D, Tw, nullptr, Q, q_mode, batch, n, k, Rw, nullptr, nullptr, nullptr, gov.status, 1, 0, st,
                                gov.marks)))
        return rc;
      if ((rc = dsge_host::launch_gensys_overlap_merge(batch, gov.marks, gov.status, status_out, logp_out, st))) return rc;
      if ((rc = launch_kalman(Tw, RQR, P0, 0, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, n, p, T_len, jitter,
                              missing_fill, n_state_hint, z_selector_hint, logp_out, status_out, st, nullptr, fold_rqr ? Rw : nullptr,
                              fold_rqr ? Q : nullptr, q_mode == DSGE_Q_DIAG_BATCHED, k, nullptr, 1)))
        return rc;
    }
    if (park_failures) {
      if ((rc = launch_status_park(status_out, (int32_t*)cm_w, batch, 1, st))) return rc;
    }
    if (ms_out) {
      HIP_TRY(hipEventRecord(ev[3], st));
      HIP_TRY(hipEventSynchronize(ev[3]));
      for (int i = 0; i < 3; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        acc_ms[i] += ms;
      }
    }
  }
  if (ms_out) {
    for (int i = 0; i < 3; ++i) ms_out[i] = acc_ms[i] / (float)n_rep;
    for (auto& e : ev) HIP_TRY(hipEventDestroy(e));
  }
  return DSGE_SUCCESS;
}

int dsge_solve_kalman_logp_batched(const double* A, const double* B, const double* C, const double* D, const double* Q,
                                   int q_mode, const double* Z, int z_batched, const double* d, int d_batched,
                                   const double* Hdiag, int h_batched, const double* y, int batch, int n, int k, int p,
                                   int T_len, int solver, double tol, int max_iter, double jitter, double missing_fill,
                                   int n_state_hint, int z_selector_hint, int n_lead_hint, double* logp_out,
                                   int32_t* status_out, double* T_out, double* R_out, double* resid_out,
                                   int32_t* n_iter_out, void* stream) {
  // Large batches run as chunks alternating over two library-owned streams (fork / join by events on the caller's
  // stream): a draw whose covariance recursion converges late -- or never, within the sample -- keeps ONE wavefront of the
  // Kalman launch busy for up to 200 full steps (1.5 ms) while the rest of the GPU has long finished; with two chunk
  // pipelines in flight that tail overlaps the cycle-reduction launch of the next chunk instead of being idle time.
  if (opt().pipeline_chunks < 2 || batch < 1024 || batch / opt().pipeline_chunks < 256)
    return pipeline(A, B, C, D, Q, q_mode, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, n, k, p, T_len, solver,
                    tol, max_iter, jitter, missing_fill, n_state_hint, z_selector_hint, n_lead_hint, logp_out, status_out,
                    T_out, R_out, resid_out, n_iter_out, (hipStream_t)stream, 1, nullptr);
  int rc = ensure_device();
  if (rc) return rc;
  constexpr int MAXS = 8;
  static thread_local hipStream_t s_str[MAXS] = {};
  static thread_local hipEvent_t s_ev[MAXS + 1] = {};
  static thread_local int s_dev = -1;
  int dev_now = 0;
  HIP_TRY(hipGetDevice(&dev_now));
  if (s_dev != dev_now) {
    for (auto& x : s_str) HIP_TRY(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (auto& e : s_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    s_dev = dev_now;
  }
  hipStream_t caller = (hipStream_t)stream;
  const int n_chunks = opt().pipeline_chunks;
  const int n_str = n_chunks < MAXS ? n_chunks : MAXS;
  const int per = ((batch + n_chunks - 1) / n_chunks + 63) & ~63;
  const size_t slice = (pipeline_scratch_bytes(per, n, k) + 255) & ~(size_t)255;
  void* base = nullptr;
  if ((rc = scratch_reserve(caller, slice * n_chunks, &base))) return rc;
  HIP_TRY(hipEventRecord(s_ev[MAXS], caller));
  for (int i = 0; i < n_str; ++i) HIP_TRY(hipStreamWaitEvent(s_str[i], s_ev[MAXS], 0));
  const bool q_b = (q_mode == DSGE_Q_DIAG_BATCHED || q_mode == DSGE_Q_FULL_BATCHED);
  const size_t qk = (q_mode == DSGE_Q_FULL_BATCHED) ? (size_t)k * k : (size_t)k;
  for (int c = 0; c * per < batch; ++c) {
    const int c0 = c * per;
    const int nb = (batch - c0 < per) ? batch - c0 : per;
    const size_t o2 = (size_t)c0 * n * n, ok = (size_t)c0 * n * k;
    rc = pipeline(A + o2, B + o2, C + o2, D + ok, q_b ? Q + c0 * qk : Q, q_mode, z_batched ? Z + (size_t)c0 * p * n : Z,
                  z_batched, (d && d_batched) ? d + (size_t)c0 * p : d, d_batched,
                  (Hdiag && h_batched) ? Hdiag + (size_t)c0 * p : Hdiag, h_batched, y, nb, n, k, p, T_len, solver, tol,
                  max_iter, jitter, missing_fill, n_state_hint, z_selector_hint, n_lead_hint, logp_out + c0, status_out + c0,
                  T_out ? T_out + o2 : nullptr, R_out ? R_out + ok : nullptr, resid_out ? resid_out + c0 : nullptr,
                  n_iter_out ? n_iter_out + c0 : nullptr, s_str[c % n_str], 1, nullptr, 0, (char*)base + slice * c);
    if (rc) break;
  }
  for (int i = 0; i < n_str; ++i) {
    HIP_TRY(hipEventRecord(s_ev[i], s_str[i]));
    HIP_TRY(hipStreamWaitEvent(caller, s_ev[i], 0));
  }
  return rc;
}

int dsge_second_order_logp_batched(const double* A, const double* B, const double* C, const double* D,
                                   const int32_t* hess_idx, int nnz, const double* hess_val, const double* q, int q_batched,
                                   const double* Z, const double* d, const double* Hdiag, const double* y, int batch, int n,
                                   int k, int p, int T_len, int solver, double tol, int max_iter, double jitter,
                                   double missing_fill, const int32_t* state_idx, int n_state, const int32_t* lead_idx,
                                   int n_lead, const int32_t* ret_idx, int n_ret, double* logp_out, int32_t* status_out,
                                   double* T_out, double* R_out, double* gyy_out, double* gyu_out, double* guu_out,
                                   double* gss_out, float* stage_ms, void* stream) {
  int rc = check_common(batch, n, DSGE_MAX_N_CR);
  if (rc) return rc;
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "k out of range (1..n)");
  if (p < 1 || p > 8) return fail(DSGE_ERR_INVALID, "second order: p out of range (1..8)");
  if (T_len < 0 || nnz < 0) return fail(DSGE_ERR_INVALID, "T_len < 0 or nnz < 0");
  if (!A || !B || !C || !D || (nnz > 0 && (!hess_idx || !hess_val)) || !q || !Z || !y || !logp_out || !status_out ||
      !state_idx || !ret_idx || (n_lead > 0 && !lead_idx))
    return fail(DSGE_ERR_INVALID, "null pointer");
  if (solver != DSGE_SOLVER_CYCLE_REDUCTION && solver != DSGE_SOLVER_GENSYS)
    return fail(DSGE_ERR_INVALID, "second order: solver must be cycle reduction or gensys");
  for (int i = 0; i < n_state; ++i)
    if (state_idx[i] < 0 || state_idx[i] >= n) return fail(DSGE_ERR_INVALID, "state_idx out of range");
  for (int i = 0; i < n_lead; ++i)
    if (lead_idx[i] < 0 || lead_idx[i] >= n) return fail(DSGE_ERR_INVALID, "lead_idx out of range");
  for (int i = 0; i < n_ret; ++i)
    if (ret_idx[i] < 0 || ret_idx[i] >= n) return fail(DSGE_ERR_INVALID, "ret_idx out of range");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  hipStream_t st = (hipStream_t)stream;
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k;
  void* base = nullptr;
  if ((rc = scratch_reserve(st, align256(nn * 8) + align256(nk * 8) + 2 * align256((size_t)batch * 12) + 4096, &base)))
    return rc;
  Carver cv(base);
  double* Tw = T_out ? T_out : cv.take<double>(nn);
  double* Rw = R_out ? R_out : cv.take<double>(nk);
  int32_t* eu_w = cv.take<int32_t>((size_t)batch * 3);
  int32_t* it_w = cv.take<int32_t>((size_t)batch);
  EventGuard e0, e1;  // (destroyed on every return path)
  if (stage_ms) {
    HIP_TRY(e0.create());
    HIP_TRY(e1.create());
    HIP_TRY(hipEventRecord(e0, st));
  }
  if (solver == DSGE_SOLVER_CYCLE_REDUCTION) {
    int deflated = 0;
    if ((rc = launch_cr_deflated(A, B, C, D, batch, n, k, max_iter, tol, Tw, Rw, status_out, it_w, st, &deflated))) return rc;
    if (!deflated && (rc = launch_cr(A, B, C, batch, n, max_iter, tol, Tw, status_out, it_w, st, 0, D, k, Rw))) return rc;
  } else {
    if ((rc = launch_gensys(A, B, C, batch, n, tol, n_lead, Tw, eu_w, status_out, st))) return rc;
    if ((rc = launch_assemble(A, B, C, D, Tw, nullptr, q, q_batched ? DSGE_Q_DIAG_BATCHED : DSGE_Q_DIAG_SHARED, batch, n, k,
                              Rw, nullptr, nullptr, nullptr, status_out, 1, 0, st)))
      return rc;
  }
  if (stage_ms) HIP_TRY(hipEventRecord(e1, st));
  rc = launch_second_order(B, C, Tw, Rw, hess_idx, nnz, hess_val, q, q_batched, Z, d, Hdiag, y, batch, n, k, p, T_len, jitter,
                           missing_fill, state_idx, n_state, lead_idx, n_lead, ret_idx, n_ret, logp_out, status_out, gyy_out,
                           gyu_out, guu_out, gss_out, g_kalman_steady_at, nullptr, st, stage_ms ? stage_ms + 1 : nullptr,
                           solver == DSGE_SOLVER_CYCLE_REDUCTION ? it_w : nullptr);
  if (rc) return rc;
  if (stage_ms) {
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipEventElapsedTime(&stage_ms[0], e0, e1));
  }
  return DSGE_SUCCESS;
}

int dsge_solve_kalman_logp_augmented_batched(const double* A, const double* B, const double* C, const double* D,
                                             const double* Q, int q_mode, const double* Z, int z_batched,
                                             const double* d, int d_batched, const double* Hdiag, int h_batched,
                                             const double* y, int batch, int n, int k, int p, int T_len, int solver,
                                             double tol, int max_iter, double jitter, double missing_fill, int m,
                                             const int32_t* inv_var_order, int n_links, const int32_t* link_rows,
                                             const int32_t* link_cols, int n_state_hint, int z_selector_hint,
                                             int n_lead_hint, double* logp_out, int32_t* status_out, double* T_aug_out,
                                             double* R_aug_out, double* resid_out, void* stream) {
  const bool is_cr = solver == DSGE_SOLVER_CYCLE_REDUCTION || solver == DSGE_SOLVER_SCAN_CYCLE_REDUCTION;
  int rc = check_common(batch, n, is_cr ? DSGE_MAX_N_CR : DSGE_MAX_N);
  if (rc) return rc;
  if (m < n || m > DSGE_MAX_N_BIG) return fail(DSGE_ERR_INVALID, "augmented state dimension m out of range (n..DSGE_MAX_N_BIG)");
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "k out of range (1..n)");
  if (p < 1 || p > DSGE_MAX_P) return fail(DSGE_ERR_INVALID, "p out of range (1..DSGE_MAX_P)");
  if (T_len < 0 || n_links < 0) return fail(DSGE_ERR_INVALID, "T_len < 0 or n_links < 0");
  if (q_mode < 0 || q_mode > 3) return fail(DSGE_ERR_INVALID, "bad q_mode");
  if (!A || !B || !C || !D || !Q || !Z || !y || !logp_out || !status_out || (n_links > 0 && (!link_rows || !link_cols)))
    return fail(DSGE_ERR_INVALID, "null pointer");
  if (!is_cr && solver != DSGE_SOLVER_BACKWARD_DIRECT && solver != DSGE_SOLVER_GENSYS)
    return fail(DSGE_ERR_INVALID, "unknown solver code");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  hipStream_t st = (hipStream_t)stream;
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k, mm = (size_t)batch * m * m, mk = (size_t)batch * m * k;
  void* base = nullptr;
  if ((rc = scratch_reserve(st, align256(nn * 8) + align256(nk * 8) + 3 * align256(mm * 8) + align256(mk * 8) +
                                         align256((size_t)batch * 12) + 4096,
                          &base)))
    return rc;
  Carver cv(base);
  double* Tw = cv.take<double>(nn);
  double* Rw = cv.take<double>(nk);
  double* Ta = T_aug_out ? T_aug_out : cv.take<double>(mm);
  double* Ra = R_aug_out ? R_aug_out : cv.take<double>(mk);
  double* RQR = cv.take<double>(mm);
  double* P0 = cv.take<double>(mm);
  int32_t* eu_w = cv.take<int32_t>((size_t)batch * 3);
  if (is_cr) {
    rc = launch_cr(A, B, C, batch, n, max_iter, tol, Tw, status_out, nullptr, st, solver == DSGE_SOLVER_SCAN_CYCLE_REDUCTION);
  } else if (solver == DSGE_SOLVER_GENSYS) {
    rc = launch_gensys(A, B, C, batch, n, tol, n_lead_hint, Tw, eu_w, status_out, st);
  } else {
    HIP_TRY(hipMemsetAsync(status_out, 0, sizeof(int32_t) * batch, st));
    rc = launch_bdirect(A, B, D, batch, n, k, Tw, Rw, st);
  }
  if (rc) return rc;
  // R and the policy residual in SOLVER order (statespace.py:213), then un-permute + augment
  if ((rc = launch_assemble(A, B, C, D, Tw, nullptr, Q, q_mode, batch, n, k, Rw, resid_out, nullptr, nullptr, status_out, 1,
                            0, st)))
    return rc;
  if ((rc = launch_augment(Tw, Rw, batch, n, k, m, inv_var_order, n_links, link_rows, link_cols, Ta, Ra, st))) return rc;
  if (m > DSGE_MAX_N) {
    // Round 6: 65 .. 96 augmented states (cumulator and observation-lag chains on a 40-variable model, statespace.py:598-723).
    // The filter does not grow with m: the model restricted to F = {non-zero columns of T_aug} u {observed variables} is exact
    // (x_t[F] depends on x_{t-1}[F] only, y_t on x_t[F] only -- the argument of the 65 .. 96-variable route, csrc/dsge_big.hpp),
    // and the existing kernels take |F| <= 64.  F is measured on the device (one 32-byte read-back: the call synchronises its
    // stream once); a model beyond 64 filtered variables is DSGE_ERR_TOO_LARGE -- with T_aug, R_aug, the status words and the
    // residual of this call already written (they do not depend on the filter).
    unsigned char idx[64];
    int u = 0, ns = 0;
    if ((rc = big_filtered_variables(Ta, Z, z_batched, batch, m, p, st, idx, &u, &ns))) return rc;
    if (u > 64)
      return fail(DSGE_ERR_TOO_LARGE, "augmented solve + Kalman with m > 64: " + std::to_string(u) +
                                          " state (incl. chain) and observed variables, the filter kernels take at most 64");
    if (u < 1) return fail(DSGE_ERR_INVALID, "augmented solve + Kalman with m > 64: no state and no observed variable");
    void* b2 = nullptr;
    const size_t uu = (size_t)batch * 64 * 64;
    if ((rc = g_aug_big_pool.reserve(3 * align256(uu * 8) + align256((size_t)batch * 64 * k * 8) +
                                         align256((size_t)batch * DSGE_MAX_P * 64 * 8) + 1024,
                                     st, &b2)))
      return rc;
    Carver c2(b2);
    double* T_r = c2.take<double>(uu);
    double* RQR_r = c2.take<double>(uu);
    double* P0_r = c2.take<double>(uu);
    double* R_r = c2.take<double>((size_t)batch * 64 * k);
    double* Z_r = c2.take<double>(z_batched ? (size_t)batch * p * 64 : (size_t)p * 64);
    if ((rc = launch_big_compress(Ta, Ra, Z, z_batched, batch, m, k, p, idx, u, T_r, R_r, Z_r, st))) return rc;
    if ((rc = launch_assemble(nullptr, nullptr, nullptr, nullptr, T_r, R_r, Q, q_mode, batch, u, k, nullptr, nullptr, RQR_r, P0_r,
                              status_out, 0, 2, st)))
      return rc;
    const int ns_hint = (ns > 0 && ns < u) ? ns : 0;
    return launch_kalman(T_r, RQR_r, P0_r, 0, Z_r, z_batched, d, d_batched, Hdiag, h_batched, y, batch, u, p, T_len, jitter,
                         missing_fill, ns_hint, z_selector_hint, logp_out, status_out, st);
  }
  if ((rc = launch_assemble(nullptr, nullptr, nullptr, nullptr, Ta, Ra, Q, q_mode, batch, m, k, nullptr, nullptr, RQR, P0,
                            status_out, 0, 2, st)))
    return rc;
  return launch_kalman(Ta, RQR, P0, 0, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, m, p, T_len, jitter,
                       missing_fill, n_state_hint, z_selector_hint, logp_out, status_out, st);
}

// dense_z = 0: the selector path.  dense_z = 1: any design matrix, by carrying the observed combinations as p extra variables
// (dsge_augment.hpp: dense_z_augment_kernel): the reverse sweep runs on the augmented model of n + p variables, the assembly
// reverse and the policy adjoints on the original n; n_filter_hint then counts the STATE variables (non-zero columns of A).
static int grad_pipeline(const double* A, const double* B, const double* C, const double* D, const double* q, int q_batched,
                         const double* Z, int z_batched, const double* d, int d_batched, const double* Hdiag, int h_batched,
                         const double* y, int batch, int n, int k, int p, int T_len, int solver, double tol, int max_iter,
                         double jitter, double missing_fill, int n_filter_hint, int n_lead_hint, double* logp_out,
                         int32_t* status_out, double* A_bar, double* B_bar, double* C_bar, double* D_bar, double* q_bar,
                         double* d_bar, double* h_bar, int dense_z, double* Z_bar, void* stream) {
  int rc = check_common(batch, n, 56);
  if (rc) return rc;
  if (dense_z && n + p > 56) return fail(DSGE_ERR_INVALID, "gradient path with a dense design matrix: n + p must not exceed 56");
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "k out of range (1..n)");
  if (p < 1 || p > 8) return fail(DSGE_ERR_INVALID, "gradient path: p out of range (1..8)");
  if (T_len < 0) return fail(DSGE_ERR_INVALID, "T_len < 0");
  if (!A || !B || !C || !D || !q || !Z || !y || !logp_out || !status_out || !A_bar || !B_bar || !C_bar || !D_bar || !q_bar)
    return fail(DSGE_ERR_INVALID, "null pointer");
  if (solver != DSGE_SOLVER_CYCLE_REDUCTION && solver != DSGE_SOLVER_GENSYS && solver != DSGE_SOLVER_SCAN_CYCLE_REDUCTION)
    return fail(DSGE_ERR_INVALID, "gradient path: solver must be cycle_reduction, scan_cycle_reduction or gensys");
  if (q_batched < 0 || q_batched > 3) return fail(DSGE_ERR_INVALID, "gradient path: q_batched is a DSGE_Q_* mode (0..3)");
  if ((rc = ensure_device())) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  hipStream_t st = (hipStream_t)stream;
  const int m = dense_z ? n + p : n;                                   // size of the model the reverse sweep runs on
  const int u_hint = dense_z ? (n_filter_hint > 0 ? n_filter_hint + p : 0) : n_filter_hint;
  const bool qfull = q_batched >= 2;                       // DSGE_Q_FULL_*: Q and q_bar are k x k
  const size_t qstride = qfull ? (size_t)k * k : (size_t)k;
  // the reverse sweep re-reads the stored (a_t, P_t): chunk the batch so that the store stays <= 16 GiB of the 288 GB
  const size_t per_draw = kalman_grad_store_doubles_per_draw(u_hint, m, T_len) * sizeof(double);
  size_t chunk = per_draw ? ((size_t)16 << 30) / per_draw : (size_t)batch;
  if (chunk < 1) chunk = 1;
  if (chunk > (size_t)batch) chunk = (size_t)batch;
  const size_t nn = chunk * n * n, nk = chunk * n * k;
  void* base = nullptr;
  const size_t mm = chunk * (size_t)m * m, mk = chunk * (size_t)m * k;
  if ((rc = scratch_reserve(st, 4 * align256(nn * 8) + align256(nk * 8) + align256(chunk * 12) + 2 * align256(chunk * 4) +
                                         align256(chunk * per_draw + 8) + 8192 +
                                         (dense_z ? 4 * align256(mm * 8) + align256(mk * 8) + align256((size_t)p * m * 8) : 0),
                          &base)))
    return rc;
  Carver cv(base);
  double* Tw = cv.take<double>(nn);
  double* Rw = cv.take<double>(nk);
  double* RQR = cv.take<double>(nn);
  double* Tbar = cv.take<double>(nn);
  double* Gbar = cv.take<double>(nn);
  int32_t* eu_w = cv.take<int32_t>(chunk * 3);
  int32_t* it_w = cv.take<int32_t>(chunk);   // cycle-reduction iterations = dispatch key of the reverse-sweep launch
  int32_t* ord_w = cv.take<int32_t>(chunk);
  double* store = cv.take<double>(chunk * per_draw / sizeof(double) + 1);
  double *Ta = nullptr, *Ra = nullptr, *RQRa = nullptr, *Tbar_a = nullptr, *Gbar_a = nullptr, *Zaug = nullptr;
  if (dense_z) {
    Ta = cv.take<double>(mm);
    Ra = cv.take<double>(mk);
    RQRa = cv.take<double>(mm);
    Tbar_a = cv.take<double>(mm);
    Gbar_a = cv.take<double>(mm);
    Zaug = cv.take<double>((size_t)p * m);
  }
  for (size_t c0 = 0; c0 < (size_t)batch; c0 += chunk) {
    const int nb = (int)(((size_t)batch - c0 < chunk) ? (size_t)batch - c0 : chunk);
    const double *Ac = A + c0 * n * n, *Bc = B + c0 * n * n, *Cc = C + c0 * n * n, *Dc = D + c0 * n * k;
    const double* qc = q + ((q_batched & 1) ? c0 * qstride : 0);
    const double* Zc = Z + (z_batched ? c0 * p * n : 0);
    const double* dc = d ? d + (d_batched ? c0 * p : 0) : nullptr;
    const double* hc = Hdiag ? Hdiag + (h_batched ? c0 * p : 0) : nullptr;
    int32_t* stc = status_out + c0;
    if (solver == DSGE_SOLVER_GENSYS)
      rc = launch_gensys(Ac, Bc, Cc, nb, n, tol, n_lead_hint, Tw, eu_w, stc, st);
    bool have_R = false;  // R from the solver's final elimination (as in the forward-only call): only sym(RQR') is left
    if (solver != DSGE_SOLVER_GENSYS) {
      int deflated = 0;  // static variables deflated first, as in the forward-only call
      if (solver == DSGE_SOLVER_CYCLE_REDUCTION &&
          (rc = launch_cr_deflated(Ac, Bc, Cc, Dc, nb, n, k, max_iter, tol, Tw, Rw, stc, it_w, st, &deflated)))
        return rc;
      have_R = deflated != 0;
      if (!deflated) {
        const bool fr = solver == DSGE_SOLVER_CYCLE_REDUCTION && opt().cr_fused_selection;
        rc = launch_cr(Ac, Bc, Cc, nb, n, max_iter, tol, Tw, stc, it_w, st, solver == DSGE_SOLVER_SCAN_CYCLE_REDUCTION,
                       fr ? Dc : nullptr, k, fr ? Rw : nullptr);
        have_R = fr;
      }
    }
    if (rc) return rc;
    if (have_R && k <= 16 && n <= 64 && !qfull) {
      if (!dense_z && (rc = launch_rqr(Rw, qc, q_batched, nb, n, k, stc, RQR, st))) return rc;
    } else if ((rc = launch_assemble(Ac, Bc, Cc, Dc, Tw, nullptr, qc, q_batched, nb, n, k, Rw, nullptr, RQR, nullptr, stc, 1, 2,
                                     st)))
      return rc;
    if (dense_z) {  // the augmented model and its sym(R Q R')
      if ((rc = launch_dense_z_augment(Tw, Rw, Zc, z_batched, nb, n, k, p, Ta, Ra, Zaug, st))) return rc;
      if ((rc = launch_assemble(nullptr, nullptr, nullptr, nullptr, Ta, Ra, qc, q_batched, nb, m, k, nullptr, nullptr, RQRa, nullptr,
                                stc, 0, 2, st)))
        return rc;
    }
    int32_t* gkey = (solver == DSGE_SOLVER_GENSYS) ? nullptr : it_w;  // (the gradient's forward sweep overwrites it with its step counts)
    if (opt().kalman_order == 0) {
      gkey = nullptr;
    } else if ((opt().kalman_order == 2 || solver == DSGE_SOLVER_GENSYS) && nb >= 512) {
      if ((rc = launch_persistence_key(Tw, stc, nb, n, it_w, st))) return rc;  // (the iteration counts are not needed again)
      gkey = it_w;
    }
    if (dense_z) {
      if ((rc = launch_kalman_grad(Ta, RQRa, Zaug, 0, dc, d_batched, hc, h_batched, y, nb, m, p, T_len, jitter, missing_fill,
                                   u_hint, store, logp_out + c0, stc, Tbar_a, Gbar_a, d_bar ? d_bar + c0 * p : nullptr,
                                   h_bar ? h_bar + c0 * p : nullptr, st, gkey, ord_w)))
        return rc;
      if ((rc = launch_dense_z_deaugment(Tbar_a, Gbar_a, Tw, RQRa, Zc, z_batched, stc, nb, n, p, Tbar, Gbar,
                                         Z_bar ? Z_bar + c0 * p * n : nullptr, st)))
        return rc;
    } else if ((rc = launch_kalman_grad(Tw, RQR, Zc, z_batched, dc, d_batched, hc, h_batched, y, nb, n, p, T_len, jitter,
                                        missing_fill, u_hint, store, logp_out + c0, stc, Tbar, Gbar,
                                        d_bar ? d_bar + c0 * p : nullptr, h_bar ? h_bar + c0 * p : nullptr, st, gkey, ord_w)))
      return rc;
    // Round 6: diagonal Q, k <= 16, up to 40 variables: one launch for the reverse of the assembly AND the policy adjoints (they
    // share the elimination of B + C T; adjoint_kernel<BS, false, true>), the two-kernel path behind it for what it flags
    if (!qfull && k <= 16 && n <= 40 && opt().grad_fused_adjoint) {
      if ((rc = launch_adjoint_fused(Bc, Cc, Tw, Rw, qc, q_batched, Gbar, Tbar, nb, n, k, A_bar + c0 * n * n, B_bar + c0 * n * n,
                                     C_bar + c0 * n * n, D_bar + c0 * n * k, q_bar + c0 * qstride, stc, st)))
        return rc;
      continue;
    }
    if ((rc = launch_grad_assemble(Bc, Cc, Tw, Rw, qc, q_batched, Gbar, nb, n, k, stc, Tbar, B_bar + c0 * n * n,
                                   C_bar + c0 * n * n, D_bar + c0 * n * k, q_bar + c0 * qstride, st)))
      return rc;
    if ((rc = launch_adjoint(Bc, Cc, Tw, Tbar, nb, n, A_bar + c0 * n * n, B_bar + c0 * n * n, C_bar + c0 * n * n, stc, st,
                             1)))
      return rc;
  }
  return DSGE_SUCCESS;
}

int dsge_solve_kalman_logp_grad_batched(const double* A, const double* B, const double* C, const double* D, const double* q,
                                        int q_batched, const double* Z, int z_batched, const double* d, int d_batched,
                                        const double* Hdiag, int h_batched, const double* y, int batch, int n, int k,
                                        int p, int T_len, int solver, double tol, int max_iter, double jitter,
                                        double missing_fill, int n_filter_hint, int n_lead_hint, double* logp_out,
                                        int32_t* status_out, double* A_bar, double* B_bar, double* C_bar, double* D_bar,
                                        double* q_bar, double* d_bar, double* h_bar, void* stream) {
  return grad_pipeline(A, B, C, D, q, q_batched, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, n, k, p, T_len, solver,
                       tol, max_iter, jitter, missing_fill, n_filter_hint, n_lead_hint, logp_out, status_out, A_bar, B_bar, C_bar,
                       D_bar, q_bar, d_bar, h_bar, 0, nullptr, stream);
}

int dsge_solve_kalman_logp_grad_dense_z_batched(const double* A, const double* B, const double* C, const double* D,
                                                const double* q, int q_batched, const double* Z, int z_batched, const double* d,
                                                int d_batched, const double* Hdiag, int h_batched, const double* y, int batch,
                                                int n, int k, int p, int T_len, int solver, double tol, int max_iter,
                                                double jitter, double missing_fill, int n_state_hint, int n_lead_hint,
                                                double* logp_out, int32_t* status_out, double* A_bar, double* B_bar,
                                                double* C_bar, double* D_bar, double* q_bar, double* d_bar, double* h_bar,
                                                double* Z_bar, void* stream) {
  return grad_pipeline(A, B, C, D, q, q_batched, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, n, k, p, T_len, solver,
                       tol, max_iter, jitter, missing_fill, n_state_hint, n_lead_hint, logp_out, status_out, A_bar, B_bar, C_bar,
                       D_bar, q_bar, d_bar, h_bar, 1, Z_bar, stream);
}

int dsge_profile_pipeline(const double* A, const double* B, const double* C, const double* D, const double* Q,
                          int q_mode, const double* Z, int z_batched, const double* d, int d_batched,
                          const double* Hdiag, int h_batched, const double* y, int batch, int n, int k, int p,
                          int T_len, int solver, double tol, int max_iter, double jitter, double missing_fill,
                          int n_state_hint, int z_selector_hint, int n_lead_hint, double* logp_out,
                          int32_t* status_out, int reps, float* ms_out, void* stream) {
  if (!ms_out || reps < 1) return fail(DSGE_ERR_INVALID, "ms_out null or reps < 1");
  return pipeline(A, B, C, D, Q, q_mode, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, n, k, p, T_len, solver,
                  tol, max_iter, jitter, missing_fill, n_state_hint, z_selector_hint, n_lead_hint, logp_out, status_out,
                  nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream, reps, ms_out);
}

// ------------------------------------------------------------------------------------------
// Host twins: stage through library-owned device buffers on the default stream.
// ------------------------------------------------------------------------------------------
#define UP(dst, src, count, type)                                                                   \
  type* dst = nullptr;                                                                              \
  if (src) {                                                                                        \
    dst = cv.take<type>(count);                                                                     \
    HIP_TRY(hipMemcpyAsync(dst, src, sizeof(type) * (count), hipMemcpyHostToDevice, tw_st));      \
  }
#define OUTBUF(dst, host, count, type) type* dst = (host) ? cv.take<type>(count) : nullptr;
#define DOWN(host, dev, count, type)                                                                \
  if (host) HIP_TRY(hipMemcpyAsync(host, dev, sizeof(type) * (count), hipMemcpyDeviceToHost, tw_st));

static int cr_host(const double* A, const double* B, const double* C, int batch, int n, int max_iter, double tol,
                   double* T_out, int32_t* status, int32_t* n_iter, int scan_mode);

int dsge_cycle_reduction_batched_host(const double* A, const double* B, const double* C, int batch, int n,
                                      int max_iter, double tol, double* T_out, int32_t* status, int32_t* n_iter) {
  return cr_host(A, B, C, batch, n, max_iter, tol, T_out, status, n_iter, 0);
}

int dsge_scan_cycle_reduction_batched_host(const double* A, const double* B, const double* C, int batch, int n,
                                           int max_iter, double tol, double* T_out, int32_t* status,
                                           int32_t* n_steps) {
  return cr_host(A, B, C, batch, n, max_iter, tol, T_out, status, n_steps, 1);
}

static int cr_host(const double* A, const double* B, const double* C, int batch, int n, int max_iter, double tol,
                   double* T_out, int32_t* status, int32_t* n_iter, int scan_mode) {
  int rc = check_common(batch, n, DSGE_MAX_N_BIG);
  if (rc) return rc;
  if (!A || !B || !C || !T_out || !status) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t nn = (size_t)batch * n * n;
  void* base = nullptr;
  STAGE_RESERVE(4 * align256(nn * 8) + 2 * align256((size_t)batch * 4) + 4096, &base);
  Carver cv(base);
  UP(dA, A, nn, double);
  UP(dB, B, nn, double);
  UP(dC, C, nn, double);
  OUTBUF(dT, T_out, nn, double);
  OUTBUF(dS, status, batch, int32_t);
  OUTBUF(dI, n_iter, batch, int32_t);
  if ((rc = cr_entry(dA, dB, dC, batch, n, max_iter, tol, dT, dS, dI, tw_st, scan_mode))) return rc;
  DOWN(T_out, dT, nn, double);
  DOWN(status, dS, batch, int32_t);
  DOWN(n_iter, dI, batch, int32_t);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

// Debug hook: when enabled, kalman_sel_kernel accumulates the shader cycles draw 0 spends in each
// of its five per-step phases, [5] = cycles in steady-state steps, [6] = number of steady-state steps,
// [7] = total; dsge_debug_kalman_phases(0/1 enable, out[8]) reads them back.
int dsge_debug_kalman_phases(int enable, long long* cycles_out) {
  int rc = ensure_device();
  if (rc) return rc;
  if (enable && !g_kalman_dbg) {
    HIP_TRY(hipMalloc((void**)&g_kalman_dbg, 16 * sizeof(long long)));
    HIP_TRY(hipMemset(g_kalman_dbg, 0, 16 * sizeof(long long)));
  }
  if (cycles_out && g_kalman_dbg) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(cycles_out, g_kalman_dbg, 16 * sizeof(long long), hipMemcpyDeviceToHost));
  }
  if (!enable && g_kalman_dbg) {
    (void)hipFree(g_kalman_dbg);
    g_kalman_dbg = nullptr;
  }
  return DSGE_SUCCESS;
}

// Debug hook: cr_big_kernel (n > 64) accumulates the shader cycles workgroup 0 spends on its first draw: [0] register-block loads,
// [1] eliminations, [2] row scatters, [3] the four products and norms, [4] iterations, [5] total (with the final solve);
// [8..12] inside the eliminations (thread 0): candidates + column, first barrier, pivot row, second barrier, update.  16 values.
int dsge_debug_big_phases(int enable, long long* cycles_out) {
  int rc = ensure_device();
  if (rc) return rc;
  if (enable && !g_big_dbg) {
    HIP_TRY(hipMalloc((void**)&g_big_dbg, 16 * sizeof(long long)));
    HIP_TRY(hipMemset(g_big_dbg, 0, 16 * sizeof(long long)));
  }
  if (cycles_out && g_big_dbg) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(cycles_out, g_big_dbg, 16 * sizeof(long long), hipMemcpyDeviceToHost));
  }
  if (!enable && g_big_dbg) {
    (void)hipFree(g_big_dbg);
    g_big_dbg = nullptr;
  }
  return DSGE_SUCCESS;
}

// Debug hook (not part of the drop-in surface): shader-clock stamps of draw 0 at the phase
// boundaries of gensys_kernel: [start, after Hessenberg-triangular, after QZ, after reordering,
// after SVDs / eu codes, end].  Device pointers; cycles_out is a HOST array of 6 int64.
int dsge_debug_gensys_phases(const double* A, const double* B, const double* C, int batch, int n, double tol,
                             int n_lead_hint, double* T_out, int32_t* eu_out, int32_t* status, long long* cycles_out) {
  int rc = check_common(batch, n, DSGE_MAX_N_GENSYS - 1);
  if (rc) return rc;
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, 6 * sizeof(long long)));
  HIP_TRY(hipMemset(d, 0, 6 * sizeof(long long)));
  rc = launch_gensys(A, B, C, batch, n, tol, n_lead_hint, T_out, eu_out, status, nullptr, d);
  if (!rc) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(cycles_out, d, 6 * sizeof(long long), hipMemcpyDeviceToHost));
  }
  (void)hipFree(d);
  return rc;
}

// Debug hook: enable = 1 allocates the stamp buffer of the window kernels (draw 0 of each launch: reduce [0..4], QZ [8..11],
// eu [16..19], post [20..26]); cycles_out (host int64[32], may be NULL) reads it back; enable = 0 frees it.
int dsge_debug_gensys_stage_ms(int enable, float* ms_out) {
  static float stage[8];
  if (ms_out)
    for (int i = 0; i < 8; ++i) ms_out[i] = stage[i];
  if (enable) {
    for (float& x : stage) x = 0.f;
    g_gensys_stage_ms = stage;
  } else {
    g_gensys_stage_ms = nullptr;
  }
  return DSGE_SUCCESS;
}

int dsge_debug_gensys_window_phases(int enable, long long* cycles_out) {
  int rc = ensure_device();
  if (rc) return rc;
  if (enable && !g_gensys_win_dbg) {
    HIP_TRY(hipMalloc((void**)&g_gensys_win_dbg, 32 * sizeof(long long)));
    HIP_TRY(hipMemset(g_gensys_win_dbg, 0, 32 * sizeof(long long)));
  }
  if (cycles_out && g_gensys_win_dbg) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(cycles_out, g_gensys_win_dbg, 32 * sizeof(long long), hipMemcpyDeviceToHost));
  }
  if (!enable && g_gensys_win_dbg) {
    (void)hipFree(g_gensys_win_dbg);
    g_gensys_win_dbg = nullptr;
  }
  return DSGE_SUCCESS;
}

int dsge_gensys_batched_host(const double* A, const double* B, const double* C, const double* D, int batch, int n,
                             int k, double tol, int n_lead_hint, double* T_out, double* R_out, int32_t* eu_out,
                             int32_t* status) {
  int rc = check_common(batch, n, (big_size(n) && opt().gensys_doubling != 0) ? DSGE_MAX_N_BIG : DSGE_MAX_N_GENSYS - 1);
  if (rc) return rc;
  if (!A || !B || !C || !T_out || !eu_out || !status) return fail(DSGE_ERR_INVALID, "null pointer");
  if (R_out && (!D || k < 1 || k > n)) return fail(DSGE_ERR_INVALID, "R_out requires D and 1 <= k <= n");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * (R_out ? k : 0);
  void* base = nullptr;
  STAGE_RESERVE(4 * align256(nn * 8) + 2 * align256(nk * 8) + 2 * align256((size_t)batch * 12) + 4096, &base);
  Carver cv(base);
  UP(dA, A, nn, double);
  UP(dB, B, nn, double);
  UP(dC, C, nn, double);
  const double* dDp = nullptr;
  if (R_out) {
    UP(dD, D, nk, double);
    dDp = dD;
  }
  OUTBUF(dT, T_out, nn, double);
  OUTBUF(dR, R_out, nk, double);
  OUTBUF(dE, eu_out, (size_t)batch * 3, int32_t);
  OUTBUF(dS, status, batch, int32_t);
  if ((rc = dsge_gensys_batched(dA, dB, dC, dDp, batch, n, k, tol, n_lead_hint, dT, dR, dE, dS, tw_st))) return rc;
  DOWN(T_out, dT, nn, double);
  DOWN(R_out, dR, nk, double);
  DOWN(eu_out, dE, (size_t)batch * 3, int32_t);
  DOWN(status, dS, batch, int32_t);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_gensys_pencil_batched_host(const double* g0, const double* g1, const double* c, const double* psi, const double* pi,
                                    int batch, int N, int k, int n_eta, double tol, double* G1_out, double* C_out,
                                    double* impact_out, double* gev_out, int32_t* eu_out, int32_t* status) {
  return dsge_gensys_pencil_full_batched_host(g0, g1, c, psi, pi, batch, N, k, n_eta, tol, G1_out, C_out, impact_out, gev_out,
                                              eu_out, status, nullptr);
}

int dsge_gensys_pencil_full_batched_host(const double* g0, const double* g1, const double* c, const double* psi,
                                         const double* pi, int batch, int N, int k, int n_eta, double tol, double* G1_out,
                                         double* C_out, double* impact_out, double* gev_out, int32_t* eu_out,
                                         int32_t* status, const dsge_gensys_forward* forward) {
  int rc = check_common(batch, N, DSGE_MAX_N_GENSYS);
  if (rc) return rc;
  if (k < 1 || n_eta < 0 || n_eta + k + 1 > 64) return fail(DSGE_ERR_INVALID, "need k >= 1, n_eta >= 0, n_eta + k + 1 <= 64");
  if (!g0 || !g1 || !psi || (n_eta > 0 && !pi) || !G1_out || !C_out || !impact_out || !gev_out || !eu_out || !status)
    return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t nn = (size_t)batch * N * N, nk = (size_t)batch * N * k, ne = (size_t)batch * N * (n_eta > 0 ? n_eta : 1),
               nv = (size_t)batch * N;
  void* base = nullptr;
  const dsge_gensys_forward fh = forward ? *forward : dsge_gensys_forward{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  STAGE_RESERVE(3 * align256(nn * 8) + 2 * align256(nk * 8) + align256(ne * 8) + 2 * align256(nv * 8) + align256(nv * 32) +
                    2 * align256((size_t)batch * 12) + (fh.f_mat ? align256(nn * 16) : 0) + (fh.y_wt ? align256(nn * 16) : 0) +
                    (fh.f_wt ? align256(nk * 16) : 0) + (fh.loose ? align256(ne * 8) : 0) + align256((size_t)batch * 4) + 8192,
                &base);
  Carver cv(base);
  UP(d0, g0, nn, double);
  UP(d1, g1, nn, double);
  UP(dc, c, nv, double);
  UP(dps, psi, nk, double);
  UP(dpi, pi, (size_t)batch * N * n_eta, double);
  OUTBUF(dG, G1_out, nn, double);
  OUTBUF(dC, C_out, nv, double);
  OUTBUF(dI, impact_out, nk, double);
  OUTBUF(dV, gev_out, nv * 4, double);
  OUTBUF(dE, eu_out, (size_t)batch * 3, int32_t);
  OUTBUF(dS, status, batch, int32_t);
  OUTBUF(dFm, fh.f_mat, nn * 2, double);
  OUTBUF(dFw, fh.f_wt, nk * 2, double);
  OUTBUF(dYw, fh.y_wt, nn * 2, double);
  OUTBUF(dLo, fh.loose, ne, double);
  OUTBUF(dNu, fh.n_unstable, batch, int32_t);
  const dsge_gensys_forward fd{dFm, dFw, dYw, dLo, dNu, fh.pi_raw};
  if ((rc = dsge_gensys_pencil_full_batched(d0, d1, dc, dps, dpi, batch, N, k, n_eta, tol, dG, dC, dI, dV, dE, dS,
                                            forward ? &fd : nullptr, tw_st)))
    return rc;
  DOWN(fh.f_mat, dFm, nn * 2, double);
  DOWN(fh.f_wt, dFw, nk * 2, double);
  DOWN(fh.y_wt, dYw, nn * 2, double);
  DOWN(fh.loose, dLo, ne, double);
  DOWN(fh.n_unstable, dNu, batch, int32_t);
  DOWN(G1_out, dG, nn, double);
  DOWN(C_out, dC, nv, double);
  DOWN(impact_out, dI, nk, double);
  DOWN(gev_out, dV, nv * 4, double);
  DOWN(eu_out, dE, (size_t)batch * 3, int32_t);
  DOWN(status, dS, batch, int32_t);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_bk_eigenvalues_batched_host(const double* A, const double* B, const double* C, int batch, int n, double tol,
                                     double* eig_re, double* eig_im, int32_t* n_eig, int32_t* n_forward,
                                     int32_t* n_unstable, int32_t* status) {
  int rc = check_common(batch, n, DSGE_MAX_N_GENSYS - 1);
  if (rc) return rc;
  if (!A || !B || !C || !eig_re || !eig_im || !n_eig || !n_forward || !n_unstable || !status)
    return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t nn = (size_t)batch * n * n, ne = (size_t)batch * 2 * n;
  void* base = nullptr;
  STAGE_RESERVE(3 * align256(nn * 8) + 2 * align256(ne * 8) + 4 * align256((size_t)batch * 4) + 4096, &base);
  Carver cv(base);
  UP(dA, A, nn, double);
  UP(dB, B, nn, double);
  UP(dC, C, nn, double);
  OUTBUF(dRe, eig_re, ne, double);
  OUTBUF(dIm, eig_im, ne, double);
  OUTBUF(dNe, n_eig, batch, int32_t);
  OUTBUF(dNf, n_forward, batch, int32_t);
  OUTBUF(dNu, n_unstable, batch, int32_t);
  OUTBUF(dS, status, batch, int32_t);
  if ((rc = dsge_bk_eigenvalues_batched(dA, dB, dC, batch, n, tol, dRe, dIm, dNe, dNf, dNu, dS, tw_st))) return rc;
  DOWN(eig_re, dRe, ne, double);
  DOWN(eig_im, dIm, ne, double);
  DOWN(n_eig, dNe, batch, int32_t);
  DOWN(n_forward, dNf, batch, int32_t);
  DOWN(n_unstable, dNu, batch, int32_t);
  DOWN(status, dS, batch, int32_t);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_selection_batched_host(const double* A, const double* B, const double* C, const double* D, const double* T,
                                int batch, int n, int k, double* R_out, double* resid_out) {
  int rc = check_common(batch, n, DSGE_MAX_N_BIG);
  if (rc) return rc;
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "k out of range (1..n)");
  if (!B || !C || !D || !T || !R_out) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k;
  void* base = nullptr;
  STAGE_RESERVE(4 * align256(nn * 8) + 2 * align256(nk * 8) + align256((size_t)batch * 8) + 4096, &base);
  Carver cv(base);
  UP(dA, A, nn, double);
  UP(dB, B, nn, double);
  UP(dC, C, nn, double);
  UP(dD, D, nk, double);
  UP(dT, T, nn, double);
  OUTBUF(dR, R_out, nk, double);
  OUTBUF(dRes, resid_out, batch, double);
  if ((rc = dsge_selection_batched(dA, dB, dC, dD, dT, batch, n, k, dR, dRes, tw_st))) return rc;
  DOWN(R_out, dR, nk, double);
  DOWN(resid_out, dRes, batch, double);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_selection_adjoints_batched_host(const double* B, const double* C, const double* T, const double* R,
                                         const double* R_bar, int batch, int n, int k, double* B_bar, double* C_bar,
                                         double* D_bar, double* T_bar) {
  int rc = check_common(batch, n, 56);
  if (rc) return rc;
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "k out of range (1..n)");
  if (!B || !C || !T || !R || !R_bar || !B_bar || !C_bar || !D_bar || !T_bar) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k;
  void* base = nullptr;
  STAGE_RESERVE(6 * align256(nn * 8) + 3 * align256(nk * 8) + 4096, &base);
  Carver cv(base);
  UP(dB, B, nn, double);
  UP(dC, C, nn, double);
  UP(dT, T, nn, double);
  UP(dR, R, nk, double);
  UP(dRb, R_bar, nk, double);
  OUTBUF(dBb, B_bar, nn, double);
  OUTBUF(dCb, C_bar, nn, double);
  OUTBUF(dDb, D_bar, nk, double);
  OUTBUF(dTb, T_bar, nn, double);
  if ((rc = dsge_selection_adjoints_batched(dB, dC, dT, dR, dRb, batch, n, k, dBb, dCb, dDb, dTb, tw_st))) return rc;
  DOWN(B_bar, dBb, nn, double);
  DOWN(C_bar, dCb, nn, double);
  DOWN(D_bar, dDb, nk, double);
  DOWN(T_bar, dTb, nn, double);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_policy_adjoints_batched_host(const double* B, const double* C, const double* T, const double* T_bar,
                                      int batch, int n, double* A_bar, double* B_bar, double* C_bar, int32_t* status) {
  int rc = check_common(batch, n, 56);
  if (rc) return rc;
  if (!B || !C || !T || !T_bar || !A_bar || !B_bar || !C_bar || !status) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t nn = (size_t)batch * n * n;
  void* base = nullptr;
  STAGE_RESERVE(7 * align256(nn * 8) + align256((size_t)batch * 4) + 4096, &base);
  Carver cv(base);
  UP(dB, B, nn, double);
  UP(dC, C, nn, double);
  UP(dT, T, nn, double);
  UP(dTb, T_bar, nn, double);
  OUTBUF(dAb, A_bar, nn, double);
  OUTBUF(dBb, B_bar, nn, double);
  OUTBUF(dCb, C_bar, nn, double);
  OUTBUF(dS, status, batch, int32_t);
  if ((rc = dsge_policy_adjoints_batched(dB, dC, dT, dTb, batch, n, dAb, dBb, dCb, dS, tw_st))) return rc;
  DOWN(A_bar, dAb, nn, double);
  DOWN(B_bar, dBb, nn, double);
  DOWN(C_bar, dCb, nn, double);
  DOWN(status, dS, batch, int32_t);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_second_order_logp_batched_host(const double* A, const double* B, const double* C, const double* D,
                                        const int32_t* hess_idx, int nnz, const double* hess_val, const double* q,
                                        int q_batched, const double* Z, const double* d, const double* Hdiag, const double* y,
                                        int batch, int n, int k, int p, int T_len, int solver, double tol, int max_iter,
                                        double jitter, double missing_fill, const int32_t* state_idx, int n_state,
                                        const int32_t* lead_idx, int n_lead, const int32_t* ret_idx, int n_ret,
                                        double* logp_out, int32_t* status_out, double* T_out, double* R_out, double* gyy_out,
                                        double* gyu_out, double* guu_out, double* gss_out) {
  int rc = check_common(batch, n, DSGE_MAX_N_CR);
  if (rc) return rc;
  if (k < 1 || k > n || p < 1 || p > 8 || T_len < 0 || nnz < 0 || n_state < 1 || n_state > 24)
    return fail(DSGE_ERR_INVALID, "second order: size out of range");
  if (!A || !B || !C || !D || (nnz > 0 && (!hess_idx || !hess_val)) || !q || !Z || !y || !logp_out || !status_out ||
      !state_idx || !ret_idx || (n_lead > 0 && !lead_idx))
    return fail(DSGE_ERR_INVALID, "null pointer");
  // the structure arguments are host arrays here as in the device entry point: validated BEFORE anything is staged
  if (n_lead < 0 || n_lead > n || n_ret < n_state || n_ret > n) return fail(DSGE_ERR_INVALID, "second order: n_lead / n_ret out of range");
  for (int i = 0; i < n_state; ++i)
    if (state_idx[i] < 0 || state_idx[i] >= n) return fail(DSGE_ERR_INVALID, "state_idx out of range");
  for (int i = 0; i < n_lead; ++i)
    if (lead_idx[i] < 0 || lead_idx[i] >= n) return fail(DSGE_ERR_INVALID, "lead_idx out of range");
  for (int i = 0; i < n_ret; ++i)
    if (ret_idx[i] < 0 || ret_idx[i] >= n) return fail(DSGE_ERR_INVALID, "ret_idx out of range");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k, nv = (size_t)batch * nnz, ss = (size_t)n_state * n_state;
  void* base = nullptr;
  STAGE_RESERVE(5 * align256(nn * 8) + 4 * align256(nk * 8) + align256(nv * 8) + align256((size_t)nnz * 12) +
                    align256((size_t)batch * k * 8) + align256(nn / n * ss * 8 + 64) + align256(nk * n_state * 8 + 64) +
                    align256((size_t)batch * n * k * k * 8 + 64) + align256((size_t)batch * n * 8) +
                    align256((size_t)p * n * 8) + 2 * align256((size_t)p * 8) + align256((size_t)T_len * p * 8 + 8) +
                    2 * align256((size_t)batch * 8) + 16384,
                &base);
  Carver cv(base);
  UP(dA, A, nn, double);
  UP(dB, B, nn, double);
  UP(dC, C, nn, double);
  UP(dD, D, nk, double);
  UP(dHi, hess_idx, (size_t)nnz * 3, int32_t);
  UP(dHv, hess_val, nv, double);
  UP(dq, q, q_batched ? (size_t)batch * k : (size_t)k, double);
  UP(dZ, Z, (size_t)p * n, double);
  UP(dd, d, p, double);
  UP(dH, Hdiag, p, double);
  UP(dy, y, (size_t)T_len * p, double);
  OUTBUF(dlp, logp_out, batch, double);
  OUTBUF(dst, status_out, batch, int32_t);
  OUTBUF(dT, T_out, nn, double);
  OUTBUF(dR, R_out, nk, double);
  OUTBUF(dgyy, gyy_out, (size_t)batch * n * ss, double);
  OUTBUF(dgyu, gyu_out, (size_t)batch * n * n_state * k, double);
  OUTBUF(dguu, guu_out, (size_t)batch * n * k * k, double);
  OUTBUF(dgss, gss_out, (size_t)batch * n, double);
  if ((rc = dsge_second_order_logp_batched(dA, dB, dC, dD, dHi, nnz, dHv, dq, q_batched, dZ, dd, dH, dy, batch, n, k, p, T_len,
                                           solver, tol, max_iter, jitter, missing_fill, state_idx, n_state, lead_idx, n_lead,
                                           ret_idx, n_ret, dlp, dst, dT, dR, dgyy, dgyu, dguu, dgss, nullptr, tw_st)))
    return rc;
  DOWN(logp_out, dlp, batch, double);
  DOWN(status_out, dst, batch, int32_t);
  DOWN(T_out, dT, nn, double);
  DOWN(R_out, dR, nk, double);
  DOWN(gyy_out, dgyy, (size_t)batch * n * ss, double);
  DOWN(gyu_out, dgyu, (size_t)batch * n * n_state * k, double);
  DOWN(guu_out, dguu, (size_t)batch * n * k * k, double);
  DOWN(gss_out, dgss, (size_t)batch * n, double);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_kalman_filter_outputs_batched_host(const double* T, const double* R, const double* Q, int q_mode, const double* Z,
                                            int z_batched, const double* d, int d_batched, const double* Hdiag, int h_batched,
                                            const double* y, int batch, int m, int k, int p, int T_len, double jitter,
                                            double missing_fill, double* ll_out, double* a_pred_out, double* a_filt_out,
                                            double* p_pred_out, double* p_filt_out, int full_cov, int32_t* status_io) {
  int rc = check_common(batch, m, DSGE_MAX_N);
  if (rc) return rc;
  if (k < 1 || k > m || p < 1 || p > DSGE_MAX_P || T_len < 0 || q_mode < 0 || q_mode > 3)
    return fail(DSGE_ERR_INVALID, "size out of range");
  if (!T || !R || !Q || !Z || !y || !ll_out || !status_io) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr;
  if ((rc = twin_streams(&tw_st, nullptr))) return rc;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t mm = (size_t)batch * m * m, mk = (size_t)batch * m * k, nq = q_elems(q_mode, batch, k);
  const size_t nz = (size_t)(z_batched ? batch : 1) * p * m, nd = (size_t)(d_batched ? batch : 1) * p,
               nh = (size_t)(h_batched ? batch : 1) * p, ny = (size_t)T_len * p, tm = (size_t)batch * T_len * m,
               tc = full_cov ? tm * m : tm;
  void* base = nullptr;
  STAGE_RESERVE(align256(mm * 8) + align256(mk * 8) + align256(nq * 8) + align256(nz * 8) + align256(nd * 8) + align256(nh * 8) +
                    align256(ny * 8) + align256((size_t)batch * T_len * 8) + 2 * align256(tm * 8) + 2 * align256(tc * 8) +
                    align256((size_t)batch * 4) + 8192,
                &base);
  Carver cv(base);
  UP(dT, T, mm, double);
  UP(dR, R, mk, double);
  UP(dQ, Q, nq, double);
  UP(dZ, Z, nz, double);
  UP(dd, d, nd, double);
  UP(dH, Hdiag, nh, double);
  UP(dy, y, ny, double);
  UP(dS, status_io, batch, int32_t);
  OUTBUF(dll, ll_out, (size_t)batch * T_len, double);
  OUTBUF(dap, a_pred_out, tm, double);
  OUTBUF(daf, a_filt_out, tm, double);
  OUTBUF(dpp, p_pred_out, tc, double);
  OUTBUF(dpf, p_filt_out, tc, double);
  if ((rc = dsge_kalman_filter_outputs_batched(dT, dR, dQ, q_mode, dZ, z_batched, dd, d_batched, dH, h_batched, dy, batch, m, k, p,
                                               T_len, jitter, missing_fill, dll, dap, daf, dpp, dpf, full_cov, dS, tw_st)))
    return rc;
  DOWN(ll_out, dll, (size_t)batch * T_len, double);
  DOWN(a_pred_out, dap, tm, double);
  DOWN(a_filt_out, daf, tm, double);
  DOWN(p_pred_out, dpp, tc, double);
  DOWN(p_filt_out, dpf, tc, double);
  DOWN(status_io, dS, batch, int32_t);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_policy_norms_batched_host(const double* A, const double* B, const double* C, const double* D, const double* T,
                                   const double* R, const int32_t* state_mask, int batch, int n, int k,
                                   double* det_norm_out, double* stoch_norm_out) {
  int rc = check_common(batch, n, 56);
  if (rc) return rc;
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "k out of range (1..n)");
  if (!A || !B || !C || !D || !T || !R || !state_mask || !det_norm_out || !stoch_norm_out)
    return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k;
  void* base = nullptr;
  STAGE_RESERVE(4 * align256(nn * 8) + 2 * align256(nk * 8) + 2 * align256((size_t)batch * 8) +
                                       align256((size_t)n * 4) + 4096, &base);
  Carver cv(base);
  UP(dA, A, nn, double);
  UP(dB, B, nn, double);
  UP(dC, C, nn, double);
  UP(dD, D, nk, double);
  UP(dT, T, nn, double);
  UP(dR, R, nk, double);
  UP(dM, state_mask, (size_t)n, int32_t);
  OUTBUF(d1, det_norm_out, batch, double);
  OUTBUF(d2, stoch_norm_out, batch, double);
  if ((rc = dsge_policy_norms_batched(dA, dB, dC, dD, dT, dR, dM, batch, n, k, d1, d2, tw_st))) return rc;
  DOWN(det_norm_out, d1, batch, double);
  DOWN(stoch_norm_out, d2, batch, double);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_backward_direct_batched_host(const double* A, const double* B, const double* D, int batch, int n, int k,
                                      double* T_out, double* R_out) {
  int rc = check_common(batch, n, DSGE_MAX_N_CR);
  if (rc) return rc;
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "k out of range (1..n)");
  if (!A || !B || !D || !T_out || !R_out) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k;
  void* base = nullptr;
  STAGE_RESERVE(3 * align256(nn * 8) + 2 * align256(nk * 8) + 4096, &base);
  Carver cv(base);
  UP(dA, A, nn, double);
  UP(dB, B, nn, double);
  UP(dD, D, nk, double);
  OUTBUF(dT, T_out, nn, double);
  OUTBUF(dR, R_out, nk, double);
  if ((rc = dsge_backward_direct_batched(dA, dB, dD, batch, n, k, dT, dR, tw_st))) return rc;
  DOWN(T_out, dT, nn, double);
  DOWN(R_out, dR, nk, double);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_lyapunov_batched_host(const double* T, const double* R, const double* Q, int q_mode, int batch, int m, int k,
                               double* P0_out, double* RQR_out, int32_t* status) {
  int rc = check_common(batch, m, DSGE_MAX_N);
  if (rc) return rc;
  if (k < 1 || k > m) return fail(DSGE_ERR_INVALID, "k out of range (1..m)");
  if (q_mode < 0 || q_mode > 3) return fail(DSGE_ERR_INVALID, "bad q_mode");
  if (!T || !R || !Q || !P0_out || !status) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t mm = (size_t)batch * m * m, mk = (size_t)batch * m * k, nq = q_elems(q_mode, batch, k);
  void* base = nullptr;
  STAGE_RESERVE(3 * align256(mm * 8) + align256(mk * 8) + align256(nq * 8) +
                                       align256((size_t)batch * 4) + 4096, &base);
  Carver cv(base);
  UP(dT, T, mm, double);
  UP(dR, R, mk, double);
  UP(dQ, Q, nq, double);
  OUTBUF(dP, P0_out, mm, double);
  OUTBUF(dX, RQR_out, mm, double);
  OUTBUF(dS, status, batch, int32_t);
  if ((rc = dsge_lyapunov_batched(dT, dR, dQ, q_mode, batch, m, k, dP, dX, dS, tw_st))) return rc;
  DOWN(P0_out, dP, mm, double);
  DOWN(RQR_out, dX, mm, double);
  DOWN(status, dS, batch, int32_t);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_solve_kalman_logp_augmented_batched_host(const double* A, const double* B, const double* C, const double* D,
                                                  const double* Q, int q_mode, const double* Z, int z_batched,
                                                  const double* d, int d_batched, const double* Hdiag, int h_batched,
                                                  const double* y, int batch, int n, int k, int p, int T_len, int solver,
                                                  double tol, int max_iter, double jitter, double missing_fill, int m,
                                                  const int32_t* inv_var_order, int n_links, const int32_t* link_rows,
                                                  const int32_t* link_cols, int n_state_hint, int z_selector_hint,
                                                  int n_lead_hint, double* logp_out, int32_t* status_out,
                                                  double* T_aug_out, double* R_aug_out, double* resid_out) {
  int rc = check_common(batch, n, DSGE_MAX_N);
  if (rc) return rc;
  if (m < n || m > DSGE_MAX_N_BIG || k < 1 || k > n || p < 1 || p > DSGE_MAX_P || T_len < 0 || n_links < 0)
    return fail(DSGE_ERR_INVALID, "bad sizes");
  if (q_mode < 0 || q_mode > 3) return fail(DSGE_ERR_INVALID, "bad q_mode");
  if (!A || !B || !C || !D || !Q || !Z || !y || !logp_out || !status_out) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k, nq = q_elems(q_mode, batch, k);
  const size_t mm = (size_t)batch * m * m, mk = (size_t)batch * m * k;
  const size_t nz = (size_t)(z_batched ? batch : 1) * p * m, nd = (size_t)(d_batched ? batch : 1) * p,
               nh = (size_t)(h_batched ? batch : 1) * p, ny = (size_t)T_len * p;
  void* base = nullptr;
  STAGE_RESERVE(3 * align256(nn * 8) + align256(nk * 8) + align256(nq * 8) + align256(nz * 8) +
                                       align256(nd * 8) + align256(nh * 8) + align256(ny * 8 + 8) + align256(mm * 8) +
                                       align256(mk * 8) + 3 * align256((size_t)batch * 8) +
                                       align256((size_t)n * 4) + 2 * align256((size_t)n_links * 4 + 4) + 8192, &base);
  Carver cv(base);
  UP(dA, A, nn, double);
  UP(dB, B, nn, double);
  UP(dC, C, nn, double);
  UP(dD, D, nk, double);
  UP(dQ, Q, nq, double);
  UP(dZ, Z, nz, double);
  UP(dd, d, nd, double);
  UP(dH, Hdiag, nh, double);
  UP(dy, y, ny, double);
  UP(dinv, inv_var_order, (size_t)n, int32_t);
  UP(dlr, link_rows, (size_t)n_links, int32_t);
  UP(dlc, link_cols, (size_t)n_links, int32_t);
  OUTBUF(dL, logp_out, batch, double);
  OUTBUF(dS, status_out, batch, int32_t);
  OUTBUF(dTa, T_aug_out, mm, double);
  OUTBUF(dRa, R_aug_out, mk, double);
  OUTBUF(dRes, resid_out, batch, double);
  if ((rc = dsge_solve_kalman_logp_augmented_batched(dA, dB, dC, dD, dQ, q_mode, dZ, z_batched, dd, d_batched, dH, h_batched,
                                                     dy, batch, n, k, p, T_len, solver, tol, max_iter, jitter, missing_fill,
                                                     m, dinv, n_links, dlr, dlc, n_state_hint, z_selector_hint, n_lead_hint,
                                                     dL, dS, dTa, dRa, dRes, tw_st)))
    return rc;
  DOWN(logp_out, dL, batch, double);
  DOWN(status_out, dS, batch, int32_t);
  DOWN(T_aug_out, dTa, mm, double);
  DOWN(R_aug_out, dRa, mk, double);
  DOWN(resid_out, dRes, batch, double);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_solve_kalman_logp_grad_batched_host(const double* A, const double* B, const double* C, const double* D,
                                             const double* q, int q_batched, const double* Z, int z_batched,
                                             const double* d, int d_batched, const double* Hdiag, int h_batched,
                                             const double* y, int batch, int n, int k, int p, int T_len, int solver,
                                             double tol, int max_iter, double jitter, double missing_fill,
                                             int n_filter_hint, int n_lead_hint, double* logp_out, int32_t* status_out,
                                             double* A_bar, double* B_bar, double* C_bar, double* D_bar, double* q_bar,
                                             double* d_bar, double* h_bar) {
  int rc = check_common(batch, n, 56);
  if (rc) return rc;
  if (k < 1 || k > n || p < 1 || p > 8 || T_len < 0) return fail(DSGE_ERR_INVALID, "bad sizes");
  if (!A || !B || !C || !D || !q || !Z || !y || !logp_out || !status_out || !A_bar || !B_bar || !C_bar || !D_bar || !q_bar)
    return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  if (q_batched < 0 || q_batched > 3) return fail(DSGE_ERR_INVALID, "gradient path: q_batched is a DSGE_Q_* mode (0..3)");
  const size_t qstride = (q_batched >= 2) ? (size_t)k * k : (size_t)k;
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k, nq = (size_t)((q_batched & 1) ? batch : 1) * qstride;
  const size_t nz = (size_t)(z_batched ? batch : 1) * p * n, nd = (size_t)(d_batched ? batch : 1) * p,
               nh = (size_t)(h_batched ? batch : 1) * p, ny = (size_t)T_len * p, bp = (size_t)batch * p;
  void* base = nullptr;
  STAGE_RESERVE(6 * align256(nn * 8) + 2 * align256(nk * 8) + align256(nq * 8) + align256(nz * 8) +
                                       align256(nd * 8) + align256(nh * 8) + align256(ny * 8 + 8) +
                                       2 * align256(bp * 8) + align256((size_t)batch * qstride * 8) +
                                       2 * align256((size_t)batch * 8) + 8192, &base);
  Carver cv(base);
  UP(dA, A, nn, double);
  UP(dB, B, nn, double);
  UP(dC, C, nn, double);
  UP(dD, D, nk, double);
  UP(dq, q, nq, double);
  UP(dZ, Z, nz, double);
  UP(dd, d, nd, double);
  UP(dH, Hdiag, nh, double);
  UP(dy, y, ny, double);
  OUTBUF(dL, logp_out, batch, double);
  OUTBUF(dS, status_out, batch, int32_t);
  OUTBUF(gA, A_bar, nn, double);
  OUTBUF(gB, B_bar, nn, double);
  OUTBUF(gC, C_bar, nn, double);
  OUTBUF(gD, D_bar, nk, double);
  OUTBUF(gq, q_bar, (size_t)batch * qstride, double);
  OUTBUF(gd, d_bar, bp, double);
  OUTBUF(gh, h_bar, bp, double);
  if ((rc = dsge_solve_kalman_logp_grad_batched(dA, dB, dC, dD, dq, q_batched, dZ, z_batched, dd, d_batched, dH, h_batched,
                                                dy, batch, n, k, p, T_len, solver, tol, max_iter, jitter, missing_fill,
                                                n_filter_hint, n_lead_hint, dL, dS, gA, gB, gC, gD, gq, gd, gh, tw_st)))
    return rc;
  DOWN(logp_out, dL, batch, double);
  DOWN(status_out, dS, batch, int32_t);
  DOWN(A_bar, gA, nn, double);
  DOWN(B_bar, gB, nn, double);
  DOWN(C_bar, gC, nn, double);
  DOWN(D_bar, gD, nk, double);
  DOWN(q_bar, gq, (size_t)batch * qstride, double);
  DOWN(d_bar, gd, bp, double);
  DOWN(h_bar, gh, bp, double);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_solve_kalman_logp_grad_dense_z_batched_host(const double* A, const double* B, const double* C, const double* D,
                                             const double* q, int q_batched, const double* Z, int z_batched,
                                             const double* d, int d_batched, const double* Hdiag, int h_batched,
                                             const double* y, int batch, int n, int k, int p, int T_len, int solver,
                                             double tol, int max_iter, double jitter, double missing_fill,
                                             int n_filter_hint, int n_lead_hint, double* logp_out, int32_t* status_out,
                                             double* A_bar, double* B_bar, double* C_bar, double* D_bar, double* q_bar,
                                             double* d_bar, double* h_bar, double* Z_bar) {
  int rc = check_common(batch, n, 56);
  if (rc) return rc;
  if (n + p > 56) return fail(DSGE_ERR_INVALID, "gradient path with a dense design matrix: n + p must not exceed 56");
  if (k < 1 || k > n || p < 1 || p > 8 || T_len < 0) return fail(DSGE_ERR_INVALID, "bad sizes");
  if (!A || !B || !C || !D || !q || !Z || !y || !logp_out || !status_out || !A_bar || !B_bar || !C_bar || !D_bar || !q_bar)
    return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  if (q_batched < 0 || q_batched > 3) return fail(DSGE_ERR_INVALID, "gradient path: q_batched is a DSGE_Q_* mode (0..3)");
  const size_t qstride = (q_batched >= 2) ? (size_t)k * k : (size_t)k;
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k, nq = (size_t)((q_batched & 1) ? batch : 1) * qstride;
  const size_t nz = (size_t)(z_batched ? batch : 1) * p * n, nd = (size_t)(d_batched ? batch : 1) * p,
               nh = (size_t)(h_batched ? batch : 1) * p, ny = (size_t)T_len * p, bp = (size_t)batch * p;
  void* base = nullptr;
  STAGE_RESERVE(6 * align256(nn * 8) + 2 * align256(nk * 8) + align256(nq * 8) + align256(nz * 8) +
                                       align256(nd * 8) + align256(nh * 8) + align256(ny * 8 + 8) +
                                       2 * align256(bp * 8) + align256((size_t)batch * qstride * 8) + align256((size_t)batch * p * n * 8) +
                                       2 * align256((size_t)batch * 8) + 8192, &base);
  Carver cv(base);
  UP(dA, A, nn, double);
  UP(dB, B, nn, double);
  UP(dC, C, nn, double);
  UP(dD, D, nk, double);
  UP(dq, q, nq, double);
  UP(dZ, Z, nz, double);
  UP(dd, d, nd, double);
  UP(dH, Hdiag, nh, double);
  UP(dy, y, ny, double);
  OUTBUF(dL, logp_out, batch, double);
  OUTBUF(dS, status_out, batch, int32_t);
  OUTBUF(gA, A_bar, nn, double);
  OUTBUF(gB, B_bar, nn, double);
  OUTBUF(gC, C_bar, nn, double);
  OUTBUF(gD, D_bar, nk, double);
  OUTBUF(gq, q_bar, (size_t)batch * qstride, double);
  OUTBUF(gd, d_bar, bp, double);
  OUTBUF(gh, h_bar, bp, double);
  OUTBUF(gZ, Z_bar, (size_t)batch * p * n, double);
  if ((rc = dsge_solve_kalman_logp_grad_dense_z_batched(dA, dB, dC, dD, dq, q_batched, dZ, z_batched, dd, d_batched, dH, h_batched,
                                                dy, batch, n, k, p, T_len, solver, tol, max_iter, jitter, missing_fill,
                                                n_filter_hint, n_lead_hint, dL, dS, gA, gB, gC, gD, gq, gd, gh, gZ, tw_st)))
    return rc;
  DOWN(logp_out, dL, batch, double);
  DOWN(status_out, dS, batch, int32_t);
  DOWN(A_bar, gA, nn, double);
  DOWN(B_bar, gB, nn, double);
  DOWN(C_bar, gC, nn, double);
  DOWN(D_bar, gD, nk, double);
  DOWN(q_bar, gq, (size_t)batch * qstride, double);
  DOWN(d_bar, gd, bp, double);
  DOWN(h_bar, gh, bp, double);
  DOWN(Z_bar, gZ, (size_t)batch * p * n, double);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_autocorrelation_batched_host(const double* T, const double* R, const double* Q, int q_mode, const double* Z,
                                      const double* Hdiag, int batch, int m, int k, int p, int n_lags, int lag_step,
                                      int correlation, double* acf_out, double* Sigma_out, int32_t* status) {
  int rc = check_common(batch, m, DSGE_MAX_N);
  if (rc) return rc;
  if (k < 1 || k > m) return fail(DSGE_ERR_INVALID, "k out of range (1..m)");
  if (q_mode < 0 || q_mode > 3) return fail(DSGE_ERR_INVALID, "bad q_mode");
  if (n_lags < 0 || lag_step < 1) return fail(DSGE_ERR_INVALID, "n_lags >= 0 and lag_step >= 1 required");
  if (Z && (p < 1 || p > DSGE_MAX_P)) return fail(DSGE_ERR_INVALID, "p out of range (1..DSGE_MAX_P)");
  if (!T || !R || !Q || !acf_out || !status) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const int dim = Z ? p : m;
  const size_t mm = (size_t)batch * m * m, mk = (size_t)batch * m * k, nq = q_elems(q_mode, batch, k);
  const size_t no = (size_t)batch * (n_lags + 1) * dim * dim;
  void* base = nullptr;
  STAGE_RESERVE(2 * align256(mm * 8) + align256(mk * 8) + align256(nq * 8) + align256(no * 8) +
                                       align256((size_t)p * m * 8) + align256((size_t)p * 8) +
                                       align256((size_t)batch * 4) + 4096, &base);
  Carver cv(base);
  UP(dT, T, mm, double);
  UP(dR, R, mk, double);
  UP(dQ, Q, nq, double);
  UP(dZ, Z, (size_t)p * m, double);
  UP(dH, Hdiag, (size_t)p, double);
  double* dSig = cv.take<double>(mm);
  OUTBUF(dO, acf_out, no, double);
  OUTBUF(dS, status, batch, int32_t);
  if ((rc = dsge_autocorrelation_batched(dT, dR, dQ, q_mode, dZ, dH, batch, m, k, p, n_lags, lag_step, correlation, dO,
                                         dSig, dS, tw_st)))
    return rc;
  DOWN(acf_out, dO, no, double);
  DOWN(Sigma_out, dSig, mm, double);
  DOWN(status, dS, batch, int32_t);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_kalman_logp_batched_host(const double* T, const double* R, const double* Q, int q_mode, const double* Z,
                                  int z_batched, const double* d, int d_batched, const double* Hdiag, int h_batched,
                                  const double* y, int batch, int m, int k, int p, int T_len, double jitter,
                                  double missing_fill, int n_state_hint, int z_selector_hint, double* logp_out,
                                  int32_t* status_io) {
  int rc = check_common(batch, m, DSGE_MAX_N);
  if (rc) return rc;
  if (k < 1 || k > m) return fail(DSGE_ERR_INVALID, "k out of range (1..m)");
  if (p < 1 || p > DSGE_MAX_P) return fail(DSGE_ERR_INVALID, "p out of range (1..DSGE_MAX_P)");
  if (T_len < 0) return fail(DSGE_ERR_INVALID, "T_len < 0");
  if (q_mode < 0 || q_mode > 3) return fail(DSGE_ERR_INVALID, "bad q_mode");
  if (!T || !R || !Q || !Z || !y || !logp_out || !status_io) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t mm = (size_t)batch * m * m, mk = (size_t)batch * m * k, nq = q_elems(q_mode, batch, k);
  const size_t nz = (size_t)(z_batched ? batch : 1) * p * m, nd = (size_t)(d_batched ? batch : 1) * p,
               nh = (size_t)(h_batched ? batch : 1) * p, ny = (size_t)T_len * p;
  void* base = nullptr;
  STAGE_RESERVE(align256(mm * 8) + align256(mk * 8) + align256(nq * 8) + align256(nz * 8) +
                              align256(nd * 8) + align256(nh * 8) + align256(ny * 8) + align256((size_t)batch * 8) +
                              align256((size_t)batch * 4) + 4096, &base);
  Carver cv(base);
  UP(dT, T, mm, double);
  UP(dR, R, mk, double);
  UP(dQ, Q, nq, double);
  UP(dZ, Z, nz, double);
  UP(dd, d, nd, double);
  UP(dH, Hdiag, nh, double);
  UP(dy, y, ny, double);
  UP(dS, status_io, batch, int32_t);
  OUTBUF(dL, logp_out, batch, double);
  if ((rc = dsge_kalman_logp_batched(dT, dR, dQ, q_mode, dZ, z_batched, dd, d_batched, dH, h_batched, dy, batch, m, k,
                                     p, T_len, jitter, missing_fill, n_state_hint, z_selector_hint, dL, dS, tw_st)))
    return rc;
  DOWN(logp_out, dL, batch, double);
  DOWN(status_io, dS, batch, int32_t);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

int dsge_solve_kalman_logp_batched_host(const double* A, const double* B, const double* C, const double* D,
                                        const double* Q, int q_mode, const double* Z, int z_batched, const double* d,
                                        int d_batched, const double* Hdiag, int h_batched, const double* y, int batch,
                                        int n, int k, int p, int T_len, int solver, double tol, int max_iter,
                                        double jitter, double missing_fill, int n_state_hint, int z_selector_hint,
                                        int n_lead_hint, double* logp_out, int32_t* status_out, double* T_out,
                                        double* R_out, double* resid_out, int32_t* n_iter_out) {
  const int solver_code = solver & ~DSGE_SOLVER_FLAG_ZERO_T_ON_FAILURE;
  int rc = check_common(batch, n, (solver_code == DSGE_SOLVER_CYCLE_REDUCTION || solver_code == DSGE_SOLVER_SCAN_CYCLE_REDUCTION ||
                                   (solver_code == DSGE_SOLVER_GENSYS && opt().gensys_doubling != 0))
                                      ? DSGE_MAX_N_BIG
                                      : DSGE_MAX_N);
  if (rc) return rc;
  if (k < 1 || k > n) return fail(DSGE_ERR_INVALID, "k out of range (1..n)");
  if (p < 1 || p > DSGE_MAX_P) return fail(DSGE_ERR_INVALID, "p out of range (1..DSGE_MAX_P)");
  if (T_len < 0) return fail(DSGE_ERR_INVALID, "T_len < 0");
  if (q_mode < 0 || q_mode > 3) return fail(DSGE_ERR_INVALID, "bad q_mode");
  if (!A || !B || !C || !D || !Q || !Z || !y || !logp_out || !status_out) return fail(DSGE_ERR_INVALID, "null pointer");
  if ((rc = ensure_device())) return rc;
  hipStream_t tw_st = nullptr, tw_st1 = nullptr;
  if ((rc = twin_streams(&tw_st, &tw_st1))) return rc;
  (void)tw_st1;
  if (batch == 0) return DSGE_SUCCESS;
  const size_t nn = (size_t)batch * n * n, nk = (size_t)batch * n * k, nq = q_elems(q_mode, batch, k);
  const size_t nz = (size_t)(z_batched ? batch : 1) * p * n, nd = (size_t)(d_batched ? batch : 1) * p,
               nh = (size_t)(h_batched ? batch : 1) * p, ny = (size_t)T_len * p;
  void* base = nullptr;
  STAGE_RESERVE(4 * align256(nn * 8) + 2 * align256(nk * 8) + align256(nq * 8) + align256(nz * 8) +
                              align256(nd * 8) + align256(nh * 8) + align256(ny * 8) + 2 * align256((size_t)batch * 8) +
                              2 * align256((size_t)batch * 4) + 8192, &base);
  Carver cv(base);
  // Shared inputs first (default stream), then the batch in chunks on two streams: while the kernels of chunk c run,
  // the host stages chunk c+1 (pageable memory: hipMemcpyAsync returns once the runtime has staged the buffer), so
  // the PCIe transfer of the Jacobians overlaps the compute.  Outputs come back in one go at the end.
  double *dA = cv.take<double>(nn), *dB = cv.take<double>(nn), *dC = cv.take<double>(nn), *dD = cv.take<double>(nk);
  const bool q_b = (q_mode == DSGE_Q_DIAG_BATCHED || q_mode == DSGE_Q_FULL_BATCHED);
  double* dQ = cv.take<double>(nq);
  double* dZ = cv.take<double>(nz);
  double* dd = d ? cv.take<double>(nd) : nullptr;
  double* dH = Hdiag ? cv.take<double>(nh) : nullptr;
  double* dy = cv.take<double>(ny);
  OUTBUF(dL, logp_out, batch, double);
  OUTBUF(dS, status_out, batch, int32_t);
  OUTBUF(dT, T_out, nn, double);
  OUTBUF(dR, R_out, nk, double);
  OUTBUF(dRes, resid_out, batch, double);
  OUTBUF(dI, n_iter_out, batch, int32_t);
  if (!q_b) HIP_TRY(hipMemcpyAsync(dQ, Q, nq * 8, hipMemcpyHostToDevice, tw_st));
  if (!z_batched) HIP_TRY(hipMemcpyAsync(dZ, Z, nz * 8, hipMemcpyHostToDevice, tw_st));
  if (d && !d_batched) HIP_TRY(hipMemcpyAsync(dd, d, nd * 8, hipMemcpyHostToDevice, tw_st));
  if (Hdiag && !h_batched) HIP_TRY(hipMemcpyAsync(dH, Hdiag, nh * 8, hipMemcpyHostToDevice, tw_st));
  HIP_TRY(hipMemcpyAsync(dy, y, ny * 8, hipMemcpyHostToDevice, tw_st));
  HIP_TRY(hipStreamSynchronize(tw_st));
  hipStream_t s_str[2] = {tw_st, tw_st1};
  const int n_chunks = (batch >= 2048) ? 4 : (batch >= 512 ? 2 : 1);
  const int per = (batch + n_chunks - 1) / n_chunks;
  const size_t qk = (q_mode == DSGE_Q_FULL_BATCHED) ? (size_t)k * k : (size_t)k;
  for (int c = 0; c < n_chunks; ++c) {
    const int c0 = c * per;
    const int nb = (batch - c0 < per) ? batch - c0 : per;
    if (nb <= 0) break;
    hipStream_t st = s_str[c & 1];
    const size_t o2 = (size_t)c0 * n * n, ok = (size_t)c0 * n * k;
    HIP_TRY(hipMemcpyAsync(dA + o2, A + o2, (size_t)nb * n * n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dB + o2, B + o2, (size_t)nb * n * n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dC + o2, C + o2, (size_t)nb * n * n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dD + ok, D + ok, (size_t)nb * n * k * 8, hipMemcpyHostToDevice, st));
    if (q_b) HIP_TRY(hipMemcpyAsync(dQ + c0 * qk, Q + c0 * qk, (size_t)nb * qk * 8, hipMemcpyHostToDevice, st));
    if (z_batched)
      HIP_TRY(hipMemcpyAsync(dZ + (size_t)c0 * p * n, Z + (size_t)c0 * p * n, (size_t)nb * p * n * 8, hipMemcpyHostToDevice, st));
    if (d && d_batched) HIP_TRY(hipMemcpyAsync(dd + (size_t)c0 * p, d + (size_t)c0 * p, (size_t)nb * p * 8, hipMemcpyHostToDevice, st));
    if (Hdiag && h_batched)
      HIP_TRY(hipMemcpyAsync(dH + (size_t)c0 * p, Hdiag + (size_t)c0 * p, (size_t)nb * p * 8, hipMemcpyHostToDevice, st));
    if ((rc = pipeline(dA + o2, dB + o2, dC + o2, dD + ok, q_b ? dQ + c0 * qk : dQ, q_mode, z_batched ? dZ + (size_t)c0 * p * n : dZ,
                       z_batched, (dd && d_batched) ? dd + (size_t)c0 * p : dd, d_batched,
                       (dH && h_batched) ? dH + (size_t)c0 * p : dH, h_batched, dy, nb, n, k, p, T_len, solver, tol, max_iter, jitter,
                       missing_fill, n_state_hint, z_selector_hint, n_lead_hint, dL + c0, dS + c0, dT ? dT + o2 : nullptr,
                       dR ? dR + ok : nullptr, dRes ? dRes + c0 : nullptr, dI ? dI + c0 : nullptr, st, 1, nullptr, c & 1)))
      return rc;
  }
  for (auto& x : s_str) HIP_TRY(hipStreamSynchronize(x));
  DOWN(logp_out, dL, batch, double);
  DOWN(status_out, dS, batch, int32_t);
  DOWN(T_out, dT, nn, double);
  DOWN(R_out, dR, nk, double);
  DOWN(resid_out, dRes, batch, double);
  DOWN(n_iter_out, dI, batch, int32_t);
  HIP_TRY(hipStreamSynchronize(tw_st));
  return DSGE_SUCCESS;
}

// ---- per-call options (include/dsge_hip.h: dsge_options) --------------------------------------------------------------
int dsge_options_init(dsge_options* o) {
  if (!o) return fail(DSGE_ERR_INVALID, "null pointer");
  const Options& d = g_defaults;
  std::memset(o, 0, sizeof(*o));
  o->struct_size = (uint32_t)sizeof(dsge_options);
  o->cr_compact = d.cr_compact;
  o->cr_fused_selection = d.cr_fused_selection;
  o->cr_deflation = d.cr_deflation;
  o->cr_two_waves = d.cr_two_waves;
  o->n_static_hint = d.n_static_hint;
  o->kalman_order = d.kalman_order;
  o->kalman_tiny = d.kalman_tiny;
  o->kalman_block = d.kalman_block;
  o->kalman_mfma = d.kalman_mfma;
  o->pipeline_chunks = d.pipeline_chunks;
  o->gensys_split = d.gensys_split;
  o->gensys_real_stage = d.gensys_real_stage;
  o->kalman_steady_tol = d.kalman_steady_tol;
  o->kalman_nt_products = d.kalman_nt_products;
  o->cr_fused_deflation = d.cr_fused_deflation;
  o->cr_four_waves = d.cr_four_waves;
  o->gensys_pairs = d.gensys_pairs;
  o->gensys_shape_cache = d.gensys_shape_cache;
  o->kalman_narrow = d.kalman_narrow;
  o->gensys_direct_blocks = d.gensys_direct_blocks;
  o->kalman_head_draws = d.kalman_head_draws;
  o->gensys_doubling = d.gensys_doubling;
  o->kalman_grad_split = d.kalman_grad_split;
  o->ll_constant = d.ll_constant;
  o->mask_d = d.mask_d;
  o->joseph = d.joseph;
  o->jitter_F = d.jitter_F;
  o->jitter_P = d.jitter_P;
  return DSGE_SUCCESS;
}

namespace {
thread_local std::vector<OptionsGuard*> t_scope;  // dsge_options_push / dsge_options_pop of this thread
}

int dsge_options_push(const dsge_options* o) {
  if (!o) return fail(DSGE_ERR_INVALID, "null pointer");
  int rc = check_options(o);
  if (rc) return rc;
  t_scope.push_back(new OptionsGuard(o));
  return DSGE_SUCCESS;
}

int dsge_options_pop(void) {
  if (t_scope.empty()) return fail(DSGE_ERR_INVALID, "dsge_options_pop without a matching push on this thread");
  delete t_scope.back();
  t_scope.pop_back();
  return DSGE_SUCCESS;
}

int dsge_solve_kalman_logp_batched_opt(const dsge_options* opt, const double* A, const double* B, const double* C,
                                       const double* D, const double* Q, int q_mode, const double* Z, int z_batched,
                                       const double* d, int d_batched, const double* Hdiag, int h_batched, const double* y,
                                       int batch, int n, int k, int p, int T_len, int solver, double tol, int max_iter,
                                       double jitter, double missing_fill, int n_state_hint, int z_selector_hint,
                                       int n_lead_hint, double* logp_out, int32_t* status_out, double* T_out, double* R_out,
                                       double* resid_out, int32_t* n_iter_out, void* stream) {
  int rc = check_options(opt);
  if (rc) return rc;
  OptionsGuard guard(opt);
  return dsge_solve_kalman_logp_batched(A, B, C, D, Q, q_mode, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, n, k, p,
                                        T_len, solver, tol, max_iter, jitter, missing_fill, n_state_hint, z_selector_hint,
                                        n_lead_hint, logp_out, status_out, T_out, R_out, resid_out, n_iter_out, stream);
}

int dsge_solve_kalman_logp_batched_host_opt(const dsge_options* opt, const double* A, const double* B, const double* C,
                                            const double* D, const double* Q, int q_mode, const double* Z, int z_batched,
                                            const double* d, int d_batched, const double* Hdiag, int h_batched,
                                            const double* y, int batch, int n, int k, int p, int T_len, int solver, double tol,
                                            int max_iter, double jitter, double missing_fill, int n_state_hint,
                                            int z_selector_hint, int n_lead_hint, double* logp_out, int32_t* status_out,
                                            double* T_out, double* R_out, double* resid_out, int32_t* n_iter_out) {
  int rc = check_options(opt);
  if (rc) return rc;
  OptionsGuard guard(opt);
  return dsge_solve_kalman_logp_batched_host(A, B, C, D, Q, q_mode, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, n,
                                             k, p, T_len, solver, tol, max_iter, jitter, missing_fill, n_state_hint,
                                             z_selector_hint, n_lead_hint, logp_out, status_out, T_out, R_out, resid_out,
                                             n_iter_out);
}

int dsge_solve_kalman_logp_grad_batched_opt(const dsge_options* opt, const double* A, const double* B, const double* C,
                                            const double* D, const double* q, int q_batched, const double* Z, int z_batched,
                                            const double* d, int d_batched, const double* Hdiag, int h_batched,
                                            const double* y, int batch, int n, int k, int p, int T_len, int solver, double tol,
                                            int max_iter, double jitter, double missing_fill, int n_filter_hint,
                                            int n_lead_hint, double* logp_out, int32_t* status_out, double* A_bar,
                                            double* B_bar, double* C_bar, double* D_bar, double* q_bar, double* d_bar,
                                            double* h_bar, void* stream) {
  int rc = check_options(opt);
  if (rc) return rc;
  OptionsGuard guard(opt);
  return dsge_solve_kalman_logp_grad_batched(A, B, C, D, q, q_batched, Z, z_batched, d, d_batched, Hdiag, h_batched, y, batch, n,
                                             k, p, T_len, solver, tol, max_iter, jitter, missing_fill, n_filter_hint,
                                             n_lead_hint, logp_out, status_out, A_bar, B_bar, C_bar, D_bar, q_bar, d_bar, h_bar,
                                             stream);
}

int dsge_solve_kalman_logp_grad_batched_host_opt(const dsge_options* opt, const double* A, const double* B, const double* C,
                                                 const double* D, const double* q, int q_batched, const double* Z,
                                                 int z_batched, const double* d, int d_batched, const double* Hdiag,
                                                 int h_batched, const double* y, int batch, int n, int k, int p, int T_len,
                                                 int solver, double tol, int max_iter, double jitter, double missing_fill,
                                                 int n_filter_hint, int n_lead_hint, double* logp_out, int32_t* status_out,
                                                 double* A_bar, double* B_bar, double* C_bar, double* D_bar, double* q_bar,
                                                 double* d_bar, double* h_bar) {
  int rc = check_options(opt);
  if (rc) return rc;
  OptionsGuard guard(opt);
  return dsge_solve_kalman_logp_grad_batched_host(A, B, C, D, q, q_batched, Z, z_batched, d, d_batched, Hdiag, h_batched, y,
                                                  batch, n, k, p, T_len, solver, tol, max_iter, jitter, missing_fill,
                                                  n_filter_hint, n_lead_hint, logp_out, status_out, A_bar, B_bar, C_bar, D_bar,
                                                  q_bar, d_bar, h_bar);
}

}  // extern "C"
