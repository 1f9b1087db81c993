// HIP-event bracketing of individual kernel launches on the stream they are launched on.
// kinds: 0..5 = conv_igemm (dtype*3 + tile-1), 8..11 = wgrad (8 + dtype*2 + (BC==128)).
#include <mutex>
#include <vector>

#include "common.h"
#include "tuning.h"
#include "profile.h"

namespace {
struct Rec { int kind; double flops, bytes, xflops; hipEvent_t a, b; int M, N, K, taps, mode, epi; };
struct ShapeAgg { int kind, M, N, K, taps, mode, epi; double launches, ms, flops, bytes, xflops; };
std::vector<ShapeAgg> g_shapes;       // filled by tf_profile_collect, read by tf_profile_shapes
std::mutex g_mu;
int g_every = 0;            // 0 = off, n = bracket every n-th profiled launch (1 = all)
unsigned long long g_seq = 0;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}  // namespace

namespace { thread_local tf::ProfScope* g_active = nullptr; }
tf::ProfScope::ProfScope(int kind, double flops, double bytes, hipStream_t s, int M, int N, int K, int taps, int mode, int epi, double exec_flops,
                         bool kernel_mode)
    : slot(-1), stream(s), state(0) {
  if (!g_every) return;
  const bool force_bracket = tf::tuning().profile_bracket;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_seq++ % (unsigned long long)g_every) return;
  Rec r{kind, flops, bytes, exec_flops < 0 ? flops : exec_flops, get_event(), get_event(), M, N, K, taps, mode, epi};
  g_recs.push_back(r);
  slot = (int)g_recs.size() - 1;
  if (kernel_mode && !force_bracket && !g_active) { state = 2; g_active = this; }
  else { (void)hipEventRecord(r.a, s); state = 1; }
}
void tf::ProfScope::begin_bracket() {
  if (state != 2) return;
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_recs[slot].a, stream);
  state = 1;
  if (g_active == this) g_active = nullptr;
}
bool tf::ProfScope::take_launch_events(hipEvent_t* a, hipEvent_t* b) {
  ProfScope* p = g_active;
  if (!p || p->state != 2) return false;
  std::lock_guard<std::mutex> lk(g_mu);
  *a = g_recs[p->slot].a; *b = g_recs[p->slot].b;
  p->state = 3;
  g_active = nullptr;
  return true;
}
void tf::ProfScope::fall_back_to_bracket() { if (g_active) g_active->begin_bracket(); }
tf::ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (state == 1) (void)hipEventRecord(g_recs[slot].b, stream);
  else if (state == 2) { g_recs[slot].kind = -1; if (g_active == this) g_active = nullptr; }      // nothing was launched under this scope: dropped at collect
}

extern "C" int tf_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_every = on < 0 ? 0 : on;
  g_seq = 0;
  return TF_OK;
}

// rows of 6 doubles: kind, launches, total_ms, algorithmic flops, algorithmic bytes, executed flops.  Blocks until the recorded events completed.
extern "C" int tf_profile_collect(double* host_out, int max_rows) {
  std::lock_guard<std::mutex> lk(g_mu);
  constexpr int NK = 24;
  double acc[NK][5] = {};
  g_shapes.clear();
  for (const Rec& r : g_recs) {
    float ms = 0.f;
    if (r.kind >= 0 && r.kind < NK && hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      acc[r.kind][0] += 1; acc[r.kind][1] += ms; acc[r.kind][2] += r.flops; acc[r.kind][3] += r.bytes; acc[r.kind][4] += r.xflops;
      ShapeAgg* hit = nullptr;
      for (ShapeAgg& q : g_shapes)
        if (q.kind == r.kind && q.M == r.M && q.N == r.N && q.K == r.K && q.taps == r.taps && q.mode == r.mode && q.epi == r.epi) { hit = &q; break; }
      if (!hit) { g_shapes.push_back(ShapeAgg{r.kind, r.M, r.N, r.K, r.taps, r.mode, r.epi, 0, 0, 0, 0, 0}); hit = &g_shapes.back(); }
      hit->launches += 1; hit->ms += ms; hit->flops += r.flops; hit->bytes += r.bytes; hit->xflops += r.xflops;
    }
    g_pool.push_back(r.a); g_pool.push_back(r.b);
  }
  g_recs.clear();
  int n = 0;
  for (int k = 0; k < NK && n < max_rows; ++k)
    if (acc[k][0] > 0) {
      host_out[n * 6 + 0] = k; host_out[n * 6 + 1] = acc[k][0]; host_out[n * 6 + 2] = acc[k][1];
      host_out[n * 6 + 3] = acc[k][2]; host_out[n * 6 + 4] = acc[k][3]; host_out[n * 6 + 5] = acc[k][4];
      ++n;
    }
  return n;
}

// per-shape view of the LAST tf_profile_collect: rows of 12 doubles (kind, M, N, K, taps, mode, epi, launches, total_ms, flops, bytes, executed flops)
extern "C" int tf_profile_shapes(double* host_out, int max_rows) {
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  for (const ShapeAgg& q : g_shapes) {
    if (n >= max_rows) break;
    double* o = host_out + (size_t)n * 12;
    o[0] = q.kind; o[1] = q.M; o[2] = q.N; o[3] = q.K; o[4] = q.taps; o[5] = q.mode; o[6] = q.epi; o[7] = q.launches; o[8] = q.ms; o[9] = q.flops; o[10] = q.bytes; o[11] = q.xflops;
    ++n;
  }
  return n;
}
