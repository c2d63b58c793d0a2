// sdfhip — C ABI implementation (include/sdfhip.h): descriptor tables, workspace carving, kernel sequencing.
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/sdfhip.h"
#include "field_inst.h"
#include "wgrad_kernels.h"
#include "optim_kernels.h"
#include "point_kernels.h"
#include "ray_kernels.h"
#include "packed_kernels.h"
#include "volsdf_render_kernels.h"
#include "grid_encode_kernels.h"
#include "theta_kernels.h"
#include "refnerf_kernels.h"

const FieldKernels* sdfhip_kernels_A();
const FieldKernels* sdfhip_kernels_B();
const FieldKernels* sdfhip_kernels_C();
const FieldKernels* sdfhip_kernels_D();
const FieldKernels* sdfhip_kernels_E();
const FieldKernels* sdfhip_kernels_W();

static thread_local char g_err[1024] = "";
void sdfhip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int sdfhip_version(void) { return 100; }
extern "C" const char* sdfhip_last_error(void) { return g_err; }
extern "C" int64_t sdfhip_padded_points(int64_t n) { return (n + 127) / 128 * 128; }


// ------------------------------------------------------------------------------------------------ kernel timing (bench)
// HIP events around selected launches, on the stream the kernel is launched on.  Disabled by default.
enum ProfSlot {
  PS_PACK = 0, PS_ENCODE, PS_GEO_FWD, PS_ASSEMBLE, PS_COL_FWD, PS_COL_BWD, PS_BWD_PREP, PS_GEO_BWD, PS_GRID_BWD, PS_WGRAD,
  PS_WREDUCE, PS_PROP_FWD, PS_PROP_BWD, PS_RENDER_FWD, PS_RENDER_BWD, PS_DENSITY_W, PS_SAMPLERS, PS_COUNT
};
static const char* kProfNames[PS_COUNT] = {
    "pack_kernel", "geo_encode_kernel", "geo_fwd_kernel", "grad_assemble_kernel", "col_fwd_kernel", "col_bwd_kernel",
    "bwd_prep_kernel", "geo_bwd_kernel", "grid_bwd_kernel", "wgrad_kernel", "wreduce_kernel", "prop_fwd_kernel",
    "prop_bwd_kernel", "neus_render_fwd_kernel", "neus_render_bwd_kernel", "density_weights_kernels", "sampler_kernels"};
// The ONE piece of process-wide mutable state in the library, and it is measurement tooling: off unless bench.py / a test switches it on.
// While off, a launch reads one atomic flag and touches nothing else.  While on, every access to the event lists happens under `mu`: the
// launches come from several threads even in a single-model process (PyTorch runs backward() on its autograd thread, the reference's
// viewer renders from a second Python thread: viewer/server/viewer_utils.py:109-135), so a thread-local state would miss half of a
// training step and an unlocked one would corrupt its vectors.  Events of different threads land in the same slots, each pair complete.
struct ProfState {
  std::atomic<bool> enabled{false};
  std::mutex mu;
  uint64_t mask = ~0ull;  // slots that record (sdfhip_profile_enable_slots)
  std::vector<hipEvent_t> start[PS_COUNT], stop[PS_COUNT];
  size_t used[PS_COUNT] = {};
};
static ProfState g_prof;
struct ProfScope {
  int slot;
  hipStream_t s;
  bool on = false;
  hipEvent_t e_stop = nullptr;
  ProfScope(int slot_, hipStream_t s_) : slot(slot_), s(s_) {
    if (!g_prof.enabled.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (!g_prof.enabled.load() || !((g_prof.mask >> slot_) & 1ull)) return;
    on = true;
    if (g_prof.used[slot] == g_prof.start[slot].size()) {
      hipEvent_t a, b;
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      g_prof.start[slot].push_back(a);
      g_prof.stop[slot].push_back(b);
    }
    const size_t i = g_prof.used[slot]++;  // the pair is claimed here: another thread's scope takes the next one
    e_stop = g_prof.stop[slot][i];
    (void)hipEventRecord(g_prof.start[slot][i], s);
  }
  ~ProfScope() {
    if (on) (void)hipEventRecord(e_stop, s);
  }
};
extern "C" int sdfhip_profile_enable(int enable) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.enabled.store(enable != 0);
  g_prof.mask = ~0ull;
  for (int i = 0; i < PS_COUNT; ++i) g_prof.used[i] = 0;
  return PS_COUNT;
}
// Events on the launches of the selected slots only (bit i = slot i).  An event pair serialises the command stream around its launch
// (the next dispatch cannot be set up under the tail of the previous kernel), ~10 - 40 us per pair on MI355X: with every launch of a
// training step instrumented that is ~1 ms per step of measurement overhead, so bench.py times the step with events on the dominant
// kernel alone and fills its per-kernel table from a separate pass.
extern "C" int sdfhip_profile_enable_slots(uint64_t slot_mask) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.enabled.store(slot_mask != 0);
  g_prof.mask = slot_mask;
  for (int i = 0; i < PS_COUNT; ++i) g_prof.used[i] = 0;
  return PS_COUNT;
}
extern "C" const char* sdfhip_profile_name(int slot) { return (slot >= 0 && slot < PS_COUNT) ? kProfNames[slot] : ""; }
// total milliseconds and number of timed launches (scopes) recorded for a slot since the last enable; waits for them.
extern "C" int sdfhip_profile_read(int slot, double* total_ms, int64_t* count) {
  SDFHIP_REQUIRE(slot >= 0 && slot < PS_COUNT && total_ms && count, "profile_read: bad argument");
  std::lock_guard<std::mutex> lk(g_prof.mu);
  double t = 0.0;
  for (size_t i = 0; i < g_prof.used[slot]; ++i) {
    float ms = 0.0f;
    SDFHIP_CHECK_HIP(hipEventSynchronize(g_prof.stop[slot][i]));
    SDFHIP_CHECK_HIP(hipEventElapsedTime(&ms, g_prof.start[slot][i], g_prof.stop[slot][i]));
    t += ms;
  }
  *total_ms = t;
  *count = (int64_t)g_prof.used[slot];
  return 0;
}

// ------------------------------------------------------------------------------------------------ grid descriptor
extern "C" int sdfhip_grid_levels(const SdfHipGridCfg* cfg, SdfHipGridLevel* levels, int64_t* n_entries) {
  SDFHIP_REQUIRE(cfg != nullptr, "grid cfg is null");
  SDFHIP_REQUIRE(cfg->n_levels >= 1 && cfg->n_levels <= kMaxLevels, "n_levels %d out of range [1,%d]", cfg->n_levels, kMaxLevels);
  SDFHIP_REQUIRE(cfg->n_features >= 2 && cfg->n_features <= 8 && cfg->n_features % 2 == 0,
                 "n_features_per_level must be 2, 4, 6 or 8 (got %d)", cfg->n_features);
  SDFHIP_REQUIRE(cfg->log2_hashmap_size >= 4 && cfg->log2_hashmap_size <= 24, "log2_hashmap_size %d unsupported", cfg->log2_hashmap_size);
  // tiny-cuda-nn GridEncoding constructor arithmetic, in fp32 (see oracle/hashgrid.py for the statement)
  // evaluated in double and rounded once (oracle/hashgrid.py make_levels: the last bit of exp2f is libm dependent)
  const double l2 = log2((double)cfg->per_level_scale);
  uint64_t off = 0;
  for (int l = 0; l < cfg->n_levels; ++l) {
    const float scale = (float)(exp2((double)l * l2) * (double)cfg->base_resolution - 1.0);
    const uint32_t res = (uint32_t)ceilf(scale) + 1u;
    uint64_t n = (uint64_t)res * res * res;
    n = (n + 7) / 8 * 8;
    const uint64_t cap = 1ull << cfg->log2_hashmap_size;
    if (n > cap) n = cap;
    if (levels != nullptr) {
      levels[l].scale = scale;
      levels[l].resolution = res;
      levels[l].size = (uint32_t)n;
      levels[l].offset = (uint32_t)off;
      levels[l].hashed = ((uint64_t)res * res * res > n) ? 1u : 0u;
    }
    off += n;
  }
  if (n_entries != nullptr) *n_entries = (int64_t)off;
  return 0;
}

static int make_grid_dev(const SdfHipGridCfg* cfg, GridDev* g) {
  SdfHipGridLevel lv[kMaxLevels];
  int64_t n = 0;
  const int rc = sdfhip_grid_levels(cfg, lv, &n);
  if (rc != 0) return rc;
  memset(g, 0, sizeof(*g));
  g->n_levels = cfg->n_levels;
  g->n_features = cfg->n_features;
  g->smoothstep = cfg->smoothstep;
  for (int l = 0; l < cfg->n_levels; ++l) {
    g->lv[l].scale = lv[l].scale;
    g->lv[l].res = lv[l].resolution;
    g->lv[l].size = lv[l].size;
    g->lv[l].offset = lv[l].offset;
    g->lv[l].hashed = lv[l].hashed;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ per-thread fork / join state
// Work that is independent of the caller's stream for a while (the atomics-bound hash-table scatter beside the weight-gradient
// GEMMs) runs on a side stream, forked from / joined back into the caller's stream with events so the caller still sees
// plain stream order.  The stream and its two events belong to the CALLING THREAD (created on first use, released at thread
// exit), not to the field handle: the handle stays immutable and can be shared by the trainer and the viewer thread
// (SURVEY section 8b), and nothing is created or destroyed on the hot path.
struct SideLane {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool tried = false;
  bool ready() {
    if (!tried) {
      tried = true;
      if (getenv("SDFHIP_NO_OVERLAP") != nullptr) return false;
      if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess) {
        release();
      }
    }
    return stream != nullptr;
  }
  void release() {
    if (fork) (void)hipEventDestroy(fork);
    if (join) (void)hipEventDestroy(join);
    if (stream) (void)hipStreamDestroy(stream);
    fork = join = nullptr;
    stream = nullptr;
  }
  ~SideLane() { release(); }
};
static thread_local SideLane g_side;

// ------------------------------------------------------------------------------------------------ "the table's gradient is enqueued" hook
// A data-parallel host wants to start the exchange of the hash table's gradient - 1.8 GB at BASELINE config 5 - the moment the scatter
// that produces it is in the queue, not when the whole backward call (scatter, then ~2 ms of weight-gradient GEMMs) returns.  The two
// field backwards call the FIELD's callback (sdfhip_field_set_table_grad_callback: state of the handle, not of the process - two models
// in one process, or a viewer thread beside the trainer, do not see each other's) right after the scatter has been enqueued on `stream`
// (the side lane when forked, else the caller's stream): a collective the host enqueues behind `stream` inside the callback runs beside
// the weight-gradient GEMMs this call enqueues next on the caller's stream.  The callee must not synchronise.
struct TableGradHook {
  std::mutex mu;
  sdfhip_table_grad_cb cb = nullptr;
  void* user = nullptr;
};

// ------------------------------------------------------------------------------------------------ field handle
struct LinearInfo {
  int out_dim, in_dim;
  int64_t w_off, b_off;
};

struct SdfHipField {
  mutable TableGradHook table_hook;
  SdfHipFieldCfg cfg;
  const FieldKernels* k;
  GridDev grid;
  int d0, n_geo, n_col, n_feat;  // in0 dims, #linear layers, grid features
  std::vector<LinearInfo> lin;   // geometry layers then colour layers
  int64_t theta_size, table_floats, packed_size;
  int64_t g_wp[kMaxLayers], g_wpT[kMaxLayers], g_bias[kMaxLayers], g_wsdf, g_bsdf;
  int64_t g_wpT_in0 = 0;  // W_skip^T restricted to the in0 columns (g_wpT[skip] holds the hidden columns only)
  int64_t c_wp[kMaxLayers], c_wpT[kMaxLayers], c_bias[kMaxLayers], c_wout, c_bout;
  int g_rowmap[kMaxLayers], g_colmap[kMaxLayers], c_rowmap[kMaxLayers], c_colmap[kMaxLayers];
  float g_scale[kMaxLayers];
  std::vector<PackDesc> pack;
  std::vector<VecDesc> vec;
  PackDesc* d_pack = nullptr;
  VecDesc* d_vec = nullptr;
  int32_t* d_maps = nullptr;
  int max_pack_elems = 0, max_vec_n = 0;
  int64_t max_partial_elems = 0, max_partial_rows = 0;
  // every split-K GEMM of a backward call with a partial region of its own (wreduce_batch_kernel): floats per split over all layers
  int64_t total_partial_elems = 0, total_partial_rows = 0;
  bool batch_wreduce = false;

  // network depth is a run-time property of the field (the kernels loop over the layers); the table k fixes the block widths
  int nl = 0, skip = -1, nlc = 0, nb3 = 0;  // hidden geometry layers, skip layer (-1: none), hidden colour layers, width below the skip
  int kb_geo(int l) const { return l == 0 ? k->nb0 : (l == skip ? nb3 + k->nb0 : k->nbh); }
  int nbo_geo(int l) const { return l == nl ? k->nbf : ((l + 1 == skip) ? nb3 : k->nbh); }
  int kb_col(int l) const { return l == 0 ? k->nbf + k->nbs : k->nbc; }
  int pe_code = 0;    // position_encoding_max_degree | (off_axis << 8): point_kernels.h pe_freqs / pe_axes / pe_cols
  int ref_flags = 0;  // kRef*: the ref-nerf options of get_colors
};

static int add_map(std::vector<int32_t>& maps, const std::vector<int32_t>& m) {
  const int off = (int)maps.size();
  maps.insert(maps.end(), m.begin(), m.end());
  return off;
}

static inline void notify_table_grad(const SdfHipField* f, const float* table_bar, hipStream_t stream) {
  sdfhip_table_grad_cb cb;
  void* user;
  {
    std::lock_guard<std::mutex> lk(f->table_hook.mu);
    cb = f->table_hook.cb;
    user = f->table_hook.user;
  }
  if (cb != nullptr) cb(user, table_bar, (sdfhip_stream_t)stream);  // outside the lock: the callee may re-register
}
extern "C" int sdfhip_field_set_table_grad_callback(SdfHipField* f, sdfhip_table_grad_cb cb, void* user) {
  SDFHIP_REQUIRE(f != nullptr, "field_set_table_grad_callback: null field");
  std::lock_guard<std::mutex> lk(f->table_hook.mu);
  f->table_hook.cb = cb;
  f->table_hook.user = cb != nullptr ? user : nullptr;
  return 0;
}

extern "C" int sdfhip_field_create(const SdfHipFieldCfg* cfg, SdfHipField** out) {
  SDFHIP_REQUIRE(cfg != nullptr && out != nullptr, "null argument");
  SdfHipField* f = new SdfHipField();
  f->cfg = *cfg;
  int64_t entries = 0;
  if (cfg->grid.n_levels == 0) {  // no grid features (NeRFField): in0 = position + encoding only; the table pointer is never read
    memset(&f->grid, 0, sizeof(f->grid));
    f->grid.n_features = 2;
  } else {
    const int rc = make_grid_dev(&cfg->grid, &f->grid);
    if (rc != 0) {
      delete f;
      return rc;
    }
    sdfhip_grid_levels(&cfg->grid, nullptr, &entries);
  }
  f->table_floats = entries * cfg->grid.n_features;
  f->n_feat = cfg->grid.n_levels * cfg->grid.n_features;
  const int H = cfg->hidden_dim, GF = cfg->geo_feat_dim, HC = cfg->hidden_dim_color, NL = cfg->num_layers,
            NLC = cfg->num_layers_color, E = cfg->appearance_dim;
  f->pe_code = cfg->pe_degree | (cfg->pe_off_axis ? kPeOffAxis : 0);
  f->ref_flags = cfg->ref_flags & (kRefDiffuse | kRefTint | kRefReflect | kRefNdotV);
  const int D0 = 3 + pe_cols(f->pe_code) + f->n_feat;
  f->d0 = D0;
  const int skip = cfg->skip_layer;
  const bool mlp_skip = cfg->skip_style == 1;  // cat([in0, h]) with every layer H wide (field_components/mlp.py) instead of the SDF field's
  auto fail = [&](const char* why) {
    sdfhip_set_error("unsupported field configuration: %s (hidden %d, layers %d, in0 %d, geo_feat %d, colour %dx%d, skip %d)", why, H,
                     NL, D0, GF, NLC, HC, skip);
    delete f;
    return -1;
  };
  if (H % 32 || GF % 32 || HC % 32) return fail("dims must be multiples of 32");
  if (skip >= 0 && (skip < 1 || skip >= NL || (!mlp_skip && H - D0 <= 0))) return fail("bad skip layer");
  if (cfg->activation != 0 && cfg->activation != 1) return fail("activation must be 0 (Softplus 100) or 1 (ReLU)");
  if (cfg->skip_style != 0 && cfg->skip_style != 1) return fail("skip_style must be 0 or 1");
  if (NL + 1 > kMaxLayers || NLC + 1 > kMaxLayers) return fail("too many layers");
  // nb3: width (blocks) of the layer below the skip concatenation.  Its H - D0 real rows are padded to the FULL hidden width,
  // so that every hidden layer has the same shape and the fused kernels can loop over them (geo_kernels.h)
  if (cfg->pe_degree < 0 || cfg->pe_degree > 16) return fail("position_encoding_max_degree out of range");
  const CsmallLayout CL = csmall_layout(f->ref_flags, E);
  const int nb0 = (D0 + 31) / 32, nb3 = skip >= 0 ? H / 32 : 0, nbs = (CL.width + 31) / 32;
  const FieldKernels* cands[] = {sdfhip_kernels_A(), sdfhip_kernels_B(), sdfhip_kernels_C(), sdfhip_kernels_D(), sdfhip_kernels_E(),
                                 sdfhip_kernels_W()};
  f->k = nullptr;
  for (const FieldKernels* k : cands) {
    // in0 may be narrower than the instantiation's in0 blocks (the encode kernel zero-fills the rest, the packed weights have zero
    // columns there): the narrowest instantiation that holds it wins
    // (likewise the colour network's small-input blocks: with use_diffuse_color the position and the gradient are not inputs, 60 columns)
    if (k->nbh == H / 32 && k->nb0 >= nb0 && k->nbf == GF / 32 && k->nbs >= nbs && k->nbc == HC / 32 && k->act == cfg->activation &&
        (f->k == nullptr || k->nb0 < f->k->nb0 || (k->nb0 == f->k->nb0 && k->nbs < f->k->nbs)))
      f->k = k;
  }
  if (f->k == nullptr) return fail("no kernel instantiation was built for this shape");
  const FieldKernels* k = f->k;
  f->nl = NL;
  f->skip = skip;
  f->nlc = NLC;
  f->nb3 = nb3;

  // ---- natural theta layout
  f->n_geo = NL + 1;
  f->n_col = NLC + 1;
  int64_t off = 0;
  for (int l = 0; l <= NL; ++l) {
    LinearInfo li;
    li.in_dim = l == 0 ? D0 : ((mlp_skip && l == skip) ? H + D0 : H);
    li.out_dim = l == NL ? 1 + GF : ((skip >= 0 && l + 1 == skip && !mlp_skip) ? H - D0 : H);
    li.w_off = off;
    off += (int64_t)li.out_dim * li.in_dim;
    li.b_off = off;
    off += li.out_dim;
    f->lin.push_back(li);
  }
  for (int l = 0; l <= NLC; ++l) {
    LinearInfo li;
    li.in_dim = l == 0 ? CL.width + GF : HC;
    li.out_dim = l == NLC ? 3 : HC;
    li.w_off = off;
    off += (int64_t)li.out_dim * li.in_dim;
    li.b_off = off;
    off += li.out_dim;
    f->lin.push_back(li);
  }
  f->theta_size = off;

  // ---- maps + pack descriptors
  std::vector<int32_t> maps;
  int64_t poff = 0;
  auto ident = [](int n, int valid, int add = 0) {
    std::vector<int32_t> m(n);
    for (int i = 0; i < n; ++i) m[i] = i < valid ? i + add : -1;
    return m;
  };
  auto add_pack = [&](int64_t src, int ld, int kb, int nbo, int rowmap, int colmap, int transpose, float scale) {
    // layer-at-a-time kernels: a workgroup owns kWideNbo of a matrix's 2 kWideNbo out-blocks - the matrix is packed as two matrices of
    // kWideNbo out-blocks, half after half (the map that is indexed by the OUT row moves on by 32 kWideNbo entries for the second)
    const int parts = (f->k->layerwise && nbo == 2 * kWideNbo) ? 2 : 1;
    const int64_t at = poff;
    for (int h = 0; h < parts; ++h) {
      PackDesc d;
      memset(&d, 0, sizeof(d));
      d.src_off = src;
      d.dst_off = poff;
      d.ld = ld;
      d.kb = kb;
      d.nbo = nbo / parts;
      d.rowmap_off = rowmap + ((!transpose) ? h * 32 * kWideNbo : 0);
      d.colmap_off = colmap + (transpose ? h * 32 * kWideNbo : 0);
      d.transpose = transpose;
      d.scale = scale;
      f->pack.push_back(d);
      poff += (int64_t)kb * d.nbo * kChunkBlockFloats;
      f->max_pack_elems = std::max(f->max_pack_elems, kb * d.nbo * 1024);
    }
    return at;
  };
  auto add_vec = [&](int64_t src, int n, int map, int stride) {
    VecDesc d;
    memset(&d, 0, sizeof(d));
    d.src_off = src;
    d.dst_off = poff;
    d.n = n;
    d.map_off = map;
    d.stride = stride;
    f->vec.push_back(d);
    const int64_t at = poff;
    poff += (n + 31) / 32 * 32;
    f->max_vec_n = std::max(f->max_vec_n, n);
    return at;
  };
  for (int l = 0; l <= NL; ++l) {
    const LinearInfo& li = f->lin[l];
    const int kb = f->kb_geo(l), nbo = f->nbo_geo(l);
    std::vector<int32_t> rowmap, colmap;
    float scale = 1.0f;
    if (l == NL) {
      rowmap = ident(nbo * 32, GF, 1);  // feature rows 1..GF of the output layer (row 0 = sdf)
      colmap = ident(kb * 32, H);
    } else {
      rowmap = ident(nbo * 32, li.out_dim);
      if (l == 0) {
        colmap = ident(kb * 32, D0);
      } else if (l == skip) {
        // layer input = cat([h, in0]) / sqrt(2)  (sdf_field.py:403-404): h occupies nb3 blocks, in0 nb0 blocks
        colmap.assign(kb * 32, -1);
        if (mlp_skip) {  // layer input = cat([in0, h]) (field_components/mlp.py:86-88), no scaling
          for (int i = 0; i < H; ++i) colmap[i] = D0 + i;
          for (int j = 0; j < D0; ++j) colmap[f->nb3 * 32 + j] = j;
        } else {
          for (int i = 0; i < H - D0; ++i) colmap[i] = i;
          for (int j = 0; j < D0; ++j) colmap[f->nb3 * 32 + j] = (H - D0) + j;
          scale = (float)(1.0 / std::sqrt(2.0));
        }
      } else {
        colmap = ident(kb * 32, li.in_dim);
      }
    }
    f->g_rowmap[l] = add_map(maps, rowmap);
    f->g_colmap[l] = add_map(maps, colmap);
    f->g_scale[l] = scale;
    f->g_wp[l] = add_pack(li.w_off, li.in_dim, kb, nbo, f->g_rowmap[l], f->g_colmap[l], 0, scale);
    if (l == skip) {
      // W_skip^T as two matrices: the columns over h (the regular hidden -> hidden shape) and the columns over in0
      const std::vector<int32_t> ch(colmap.begin(), colmap.begin() + f->nb3 * 32), c0(colmap.begin() + f->nb3 * 32, colmap.end());
      f->g_wpT[l] = add_pack(li.w_off, li.in_dim, nbo, f->nb3, f->g_rowmap[l], add_map(maps, ch), 1, scale);
      f->g_wpT_in0 = add_pack(li.w_off, li.in_dim, nbo, k->nb0, f->g_rowmap[l], add_map(maps, c0), 1, scale);
    } else {
      f->g_wpT[l] = add_pack(li.w_off, li.in_dim, nbo, kb, f->g_rowmap[l], f->g_colmap[l], 1, scale);
    }
    f->g_bias[l] = add_vec(li.b_off, nbo * 32, f->g_rowmap[l], 1);
  }
  {
    const LinearInfo& li = f->lin[NL];
    const int m = add_map(maps, ident(k->nbh * 32, H));
    f->g_wsdf = add_vec(li.w_off, k->nbh * 32, m, 1);
    const int m1 = add_map(maps, ident(1, 1));
    f->g_bsdf = add_vec(li.b_off, 1, m1, 1);
  }
  for (int l = 0; l < NLC; ++l) {
    const LinearInfo& li = f->lin[f->n_geo + l];
    const int kb = f->kb_col(l), nbo = k->nbc;
    std::vector<int32_t> rowmap = ident(nbo * 32, HC), colmap;
    if (l == 0) {
      // reference column order (sdf_field.py:566-583): [x(3)] D(27) [grad(3)] feat(GF) emb(E) [n.v]; ours: [feat | small inputs in the
      // same order, csmall_layout]: the small columns in front of the embedding precede the feature in the reference, the rest follow it
      colmap.assign(kb * 32, -1);
      for (int i = 0; i < GF; ++i) colmap[i] = CL.emb + i;
      for (int j = 0; j < CL.width; ++j) colmap[k->nbf * 32 + j] = j < CL.emb ? j : GF + j;
    } else {
      colmap = ident(kb * 32, HC);
    }
    f->c_rowmap[l] = add_map(maps, rowmap);
    f->c_colmap[l] = add_map(maps, colmap);
    f->c_wp[l] = add_pack(li.w_off, li.in_dim, kb, nbo, f->c_rowmap[l], f->c_colmap[l], 0, 1.0f);
    f->c_wpT[l] = add_pack(li.w_off, li.in_dim, nbo, kb, f->c_rowmap[l], f->c_colmap[l], 1, 1.0f);
    f->c_bias[l] = add_vec(li.b_off, nbo * 32, f->c_rowmap[l], 1);
  }
  {
    const LinearInfo& li = f->lin[f->n_geo + NLC];
    f->c_rowmap[NLC] = add_map(maps, ident(32, 3));
    f->c_colmap[NLC] = add_map(maps, ident(k->nbc * 32, HC));
    f->c_wout = poff;
    for (int c = 0; c < 3; ++c) add_vec(li.w_off + (int64_t)c * HC, k->nbc * 32, f->c_colmap[NLC], 1);
    const int m3 = add_map(maps, ident(3, 3));
    f->c_bout = add_vec(li.b_off, 3, m3, 1);
  }
  f->packed_size = poff + 16384;  // slack behind the last chunk: the DMA moves whole 4 KiB rounds, and a layer run from the run-time layer loop prefetches its successor's first chunk at the LARGEST size any successor has (geo_kernels.h)
  // largest split-K partial
  auto upd = [&](int rows_blocks, int col_blocks) {
    f->max_partial_elems = std::max<int64_t>(f->max_partial_elems, (int64_t)rows_blocks * 32 * col_blocks * 32);
    f->max_partial_rows = std::max<int64_t>(f->max_partial_rows, (int64_t)rows_blocks * 32);
  };
  for (int l = 0; l <= NL; ++l) upd(f->nbo_geo(l), f->kb_geo(l));
  for (int l = 0; l < NLC; ++l) upd(k->nbc, f->kb_col(l));
  upd(1, k->nbc);
  f->max_partial_elems = std::max<int64_t>(f->max_partial_elems, k->nbh * 32 + 32);
  {
    auto add = [&](int rows_blocks, int col_blocks) {
      f->total_partial_elems += (int64_t)rows_blocks * 32 * col_blocks * 32;
      f->total_partial_rows += (int64_t)rows_blocks * 32;
    };
    for (int l = 0; l <= NL; ++l) add(f->nbo_geo(l), f->kb_geo(l));
    for (int l = 0; l < NLC; ++l) add(k->nbc, f->kb_col(l));
    add(1, k->nbc);
    // one region per GEMM while that stays a modest part of the training workspace (256-wide networks: ~0.9 GB at 256 splits; the
    // 512-wide layer-at-a-time path keeps the one shared buffer); SDFHIP_WREDUCE_PER_GEMM=1: the per-GEMM reductions of rounds 1 - 6 (A/B)
    static const bool per_gemm = [] { const char* e = getenv("SDFHIP_WREDUCE_PER_GEMM"); return e != nullptr && e[0] == '1'; }();
    f->batch_wreduce = !per_gemm && (int)f->lin.size() <= kWreduceBatchMax && f->total_partial_elems * 256 * 4 <= (int64_t)2 << 30;
  }

  hipError_t e = hipMalloc((void**)&f->d_pack, f->pack.size() * sizeof(PackDesc));
  if (e == hipSuccess) e = hipMalloc((void**)&f->d_vec, f->vec.size() * sizeof(VecDesc));
  if (e == hipSuccess) e = hipMalloc((void**)&f->d_maps, maps.size() * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemcpy(f->d_pack, f->pack.data(), f->pack.size() * sizeof(PackDesc), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(f->d_vec, f->vec.data(), f->vec.size() * sizeof(VecDesc), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(f->d_maps, maps.data(), maps.size() * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    sdfhip_set_error("field_create: device table upload failed: %s", hipGetErrorString(e));
    sdfhip_field_destroy(f);
    return -2;
  }
  *out = f;
  return 0;
}

extern "C" void sdfhip_field_destroy(SdfHipField* f) {
  if (f == nullptr) return;
  if (f->d_pack) (void)hipFree(f->d_pack);
  if (f->d_vec) (void)hipFree(f->d_vec);
  if (f->d_maps) (void)hipFree(f->d_maps);
  delete f;
}

extern "C" int64_t sdfhip_field_theta_size(const SdfHipField* f) { return f->theta_size; }
extern "C" int32_t sdfhip_field_num_linear(const SdfHipField* f) { return (int32_t)f->lin.size(); }
extern "C" int sdfhip_field_theta_layout(const SdfHipField* f, int64_t* w_off, int64_t* b_off, int32_t* out_dim, int32_t* in_dim) {
  for (size_t i = 0; i < f->lin.size(); ++i) {
    w_off[i] = f->lin[i].w_off;
    b_off[i] = f->lin[i].b_off;
    out_dim[i] = f->lin[i].out_dim;
    in_dim[i] = f->lin[i].in_dim;
  }
  return 0;
}
extern "C" int64_t sdfhip_field_table_size(const SdfHipField* f) { return f->table_floats; }
extern "C" int64_t sdfhip_field_packed_size(const SdfHipField* f) { return f->packed_size; }

// ---- workspace carving
struct FieldWs {
  float *x, *in0, *dydp, *u[kMaxLayers], *r[kMaxLayers], *feat, *e, *csmall, *h[kMaxLayers];
  float *rgb;
  float *gtot, *ebar, *sdfbar, *qb[kMaxLayers + 1], *zb[kMaxLayers], *in0bar, *d[kMaxLayers], *dout, *featbar, *csmallbar;
  float *partial, *bpartial;
  float* sdfrow_partial;  // the sdf row's own split partials (behind the GEMMs' regions when those are batched)
  struct WreduceBatch* batch;  // non-null: run_wgrad defers its reduction to flush_wreduce (set by the backward entry points)
  float *gsave, *dsave;  // ref-nerf options: d sdf / dx and the ray direction per point, for the backward (null otherwise)
  int n_split;
  size_t bytes;
};

// split-K partial buffers of a training workspace: one region per GEMM + the sdf row's (batched reductions) or the largest GEMM's (shared)
struct WreduceBatch {
  WreduceBatchArgs args;
  int64_t used = 0, bused = 0, cap = 0, bcap = 0;
  WreduceBatch() { args.n = 0; }
};
static int64_t partial_floats_per_split(const SdfHipField* f) {
  return f->batch_wreduce ? f->total_partial_elems + (f->k->nbh * 32 + 32) : f->max_partial_elems;
}
static int64_t bpartial_floats_per_split(const SdfHipField* f) { return f->batch_wreduce ? f->total_partial_rows : f->max_partial_rows; }
static float* sdfrow_region(const SdfHipField* f, float* partial, int n_split) {
  return (partial != nullptr && f->batch_wreduce) ? partial + (int64_t)n_split * f->total_partial_elems : partial;
}


// level: 0 = point modes (sdf / geonetwork inference), 1 = MODE_FULL with a backward to follow (every saved tensor, gradient
// staging, split-K partials), 2 = MODE_FULL forward only (z_l for the analytic-normal chain; no r_l, h_l or gradient buffers)
static void carve(const SdfHipField* f, int64_t n_points, int level, void* base, FieldWs* w) {
  const int full = level != 0, train = level == 1;
  const FieldKernels* k = f->k;
  const int64_t np = sdfhip_padded_points(n_points);
  size_t off = 0;
  auto take = [&](int64_t floats) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += ((size_t)floats * sizeof(float) + 255) / 256 * 256;
    return p;
  };
  memset(w, 0, sizeof(*w));
  w->x = take(np * 3);
  w->in0 = take(np * k->nb0 * 32);
  w->feat = take(np * k->nbf * 32);
  if (!full && k->layerwise)  // point modes of the layer-at-a-time kernels: the activations are the hand-over between launches
    for (int l = 0; l < f->nl; ++l) w->u[l] = take(np * f->nbo_geo(l) * 32);
  if (full) {
    w->dydp = take(np * f->n_feat * 3);
    for (int l = 0; l < f->nl; ++l) {
      w->u[l] = take(np * f->nbo_geo(l) * 32);
      if (train || k->layerwise) w->r[l] = take(np * f->nbo_geo(l) * 32);
    }
    w->e = take(np * k->nb0 * 32);
    w->csmall = take(np * k->nbs * 32);
    w->rgb = take(np * 3);
  }
  if (train) {
    for (int l = 0; l < f->nlc; ++l) w->h[l] = take(np * k->nbc * 32);
    w->gtot = take(np * 3);
    w->ebar = take(np * k->nb0 * 32);
    w->sdfbar = take(np);
    for (int l = 1; l <= f->nl; ++l) w->qb[l] = take(np * f->kb_geo(l) * 32);
    for (int l = 0; l < f->nl; ++l) w->zb[l] = take(np * f->nbo_geo(l) * 32);
    w->in0bar = take(np * k->nb0 * 32);
    for (int l = 0; l < f->nlc; ++l) w->d[l] = take(np * k->nbc * 32);
    w->dout = take(np * 32);
    w->featbar = take(np * k->nbf * 32);
    w->csmallbar = take(np * k->nbs * 32);
    const int64_t n_tiles = np / 32;
    w->n_split = (int)std::min<int64_t>(256, n_tiles);
    w->partial = take((int64_t)w->n_split * partial_floats_per_split(f));
    w->bpartial = take((int64_t)w->n_split * bpartial_floats_per_split(f));
    w->sdfrow_partial = sdfrow_region(f, w->partial, w->n_split);
    if (f->ref_flags & (kRefReflect | kRefNdotV)) {  // behind everything else: the layout without the options is what it always was
      w->gsave = take(np * 3);
      w->dsave = take(np * 3);
    }
  }
  w->bytes = off;
}

extern "C" int64_t sdfhip_field_workspace_size(const SdfHipField* f, int64_t n_points, int32_t training) {
  FieldWs w;
  carve(f, n_points, training, nullptr, &w);
  return (int64_t)w.bytes;
}

extern "C" int sdfhip_field_pack(const SdfHipField* f, const float* theta, float* packed, sdfhip_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  dim3 g1((f->max_pack_elems + 255) / 256, (unsigned)f->pack.size());
  { ProfScope ps_(PS_PACK, s); pack_kernel<<<g1, 256, 0, s>>>(theta, f->d_pack, f->d_maps, packed);
  dim3 g2((f->max_vec_n + 255) / 256, (unsigned)f->vec.size());
  packvec_kernel<<<g2, 256, 0, s>>>(theta, f->d_vec, f->d_maps, packed); }
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// ns: the precision mode of the kernel that will read the weights (mlp_core.h): the pointers address the first part it streams
static void fill_geo_ptrs(const SdfHipField* f, const float* packed, GeoPtrs* p, const int ns) {
  memset(p, 0, sizeof(*p));
  p->nl = f->nl;
  p->skip = f->skip;
  for (int l = 0; l <= f->nl; ++l) {
    // (layer-at-a-time kernels: 16-out-block matrices are packed as two halves of kWideNbo, see add_pack in sdfhip_field_create)
    auto chunk_nbo = [&](int nbo) { return (f->k->layerwise && nbo == 2 * kWideNbo) ? kWideNbo : nbo; };
    p->wp[l] = packed + f->g_wp[l] + chunk_part_offset(chunk_nbo(f->nbo_geo(l)), ns);
    p->wpT[l] = packed + f->g_wpT[l] + chunk_part_offset(chunk_nbo(l == f->skip ? f->nb3 : f->kb_geo(l)), ns);
    p->bias[l] = packed + f->g_bias[l];
  }
  if (f->skip >= 0) p->wpT_in0 = packed + f->g_wpT_in0 + chunk_part_offset(f->k->nb0, ns);
  p->w_sdf = packed + f->g_wsdf;
  p->b_sdf = packed + f->g_bsdf;
}
static void fill_col_ptrs(const SdfHipField* f, const float* packed, ColPtrs* p, const int ns) {
  memset(p, 0, sizeof(*p));
  p->nlc = f->nlc;
  for (int l = 0; l < f->nlc; ++l) {
    p->wp[l] = packed + f->c_wp[l] + chunk_part_offset(f->k->nbc, ns);
    p->wpT[l] = packed + f->c_wpT[l] + chunk_part_offset(f->kb_col(l), ns);
    p->bias[l] = packed + f->c_bias[l];
  }
  p->w_out = packed + f->c_wout;
  p->b_out = packed + f->c_bout;
  // use_diffuse_color: the network's sigmoid is the SPECULAR term, combined with the diffuse head and padded afterwards
  // (sdfhip_refnerf_forward): the colour kernels then return (and differentiate) the bare sigmoid
  p->rgb_padding = (f->ref_flags & kRefDiffuse) ? 0.0f : f->cfg.rgb_padding;
}

__global__ void untp_kernel(const float* __restrict__ tp, const int nb, const int n_feat, const int64_t n_rows, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * n_feat) return;
  const int64_t p = idx / n_feat;
  const int c = (int)(idx % n_feat);
  out[idx] = tp[tp_index(p, c, nb)];
}

extern "C" int sdfhip_field_forward(const SdfHipField* f, const float* packed, const float* table, const float* level_mask,
                                    const float* origins, const float* dirs, const float* starts, int64_t n_rays, int32_t n_samples,
                                    const float* emb, int32_t mode, int32_t training, void* workspace, float* sdf, float* grad,
                                    float* rgb, float* feat, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(f && packed && table && level_mask && origins && workspace && sdf, "field_forward: null argument");
  SDFHIP_REQUIRE(mode >= SDFHIP_MODE_SDF && mode <= SDFHIP_MODE_FULL, "field_forward: bad mode %d", mode);
  SDFHIP_REQUIRE(n_samples >= 1 && n_rays >= 0, "field_forward: bad shape");
  SDFHIP_REQUIRE(mode != SDFHIP_MODE_FULL || (dirs && starts && grad && rgb), "field_forward: MODE_FULL needs dirs, starts, grad, rgb");
  SDFHIP_REQUIRE(mode != SDFHIP_MODE_FULL || f->k->geo_bwd != nullptr,
                 "field_forward: MODE_FULL (analytic normal, second-order backward) exists for Softplus networks only; ReLU background fields "
                 "go through sdfhip_geo_forward / sdfhip_color_forward");
  if (n_rays == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const FieldKernels* k = f->k;
  const int64_t P = n_rays * n_samples, NP = sdfhip_padded_points(P);
  const int full = mode == SDFHIP_MODE_FULL;
  FieldWs w;
  const int save = full && training != 0;  // MODE_FULL without a backward to follow: nothing is saved (workspace level 2)
  carve(f, P, full ? (save ? 1 : 2) : 0, workspace, &w);

  EncodeArgs ea;
  memset(&ea, 0, sizeof(ea));
  ea.grid = f->grid;
  ea.origins = origins;
  ea.dirs = dirs;
  ea.starts = starts;
  ea.n_points = P;
  ea.n_padded = NP;
  ea.S = n_samples;
  // get_outputs contracts the sample positions (sdf_field.py:629); get_sdf / forward_geonetwork do NOT (:412-418, :380)
  ea.contract = full ? f->cfg.contract : 0;  // 0 none, 1 L-inf, 2 L2
  ea.pe_degree = f->pe_code;
  ea.use_pe = f->cfg.use_position_encoding;
  ea.nb0 = k->nb0;
  ea.table = table;
  ea.mask = level_mask;
  ea.x_out = w.x;
  ea.in0_tp = w.in0;
  ea.dydp = full ? w.dydp : nullptr;
  {
    ProfScope ps_(PS_ENCODE, s);
    const unsigned gx = (unsigned)(NP / 256 + (NP % 256 != 0));
    if (f->grid.n_features == 8) geo_encode8_kernel<<<dim3(gx, f->grid.n_levels + 1), 256, 0, s>>>(ea);
    else geo_encode_kernel<<<dim3(gx, f->grid.n_levels * (f->grid.n_features / 2) + 1), 256, 0, s>>>(ea);
  }

  GeoFwdArgs ga;
  memset(&ga, 0, sizeof(ga));
  fill_geo_ptrs(f, packed, &ga.p, kNsFwd);
  ga.in0_tp = w.in0;
  for (int l = 0; l < f->nl; ++l) {
    ga.u_tp[l] = w.u[l];
    ga.r_tp[l] = w.r[l];
  }
  ga.feat_tp = w.feat;
  ga.sdf = sdf;
  ga.e_tp = w.e;
  const unsigned grid = (unsigned)(NP / 128);
  { ProfScope ps_(PS_GEO_FWD, s); k->geo_fwd(full ? (save ? 0 : 4) : (mode == SDFHIP_MODE_GEO ? 1 : 2), ga, grid, s); }

  if (full) {
    AssembleArgs aa;
    memset(&aa, 0, sizeof(aa));
    aa.e_tp = w.e;
    aa.x = w.x;
    aa.dydp = w.dydp;
    aa.mask = level_mask;
    aa.dirs = dirs;
    aa.emb = emb;
    aa.n_points = P;
    aa.n_padded = NP;
    aa.S = n_samples;
    aa.pe_degree = f->pe_code;
    aa.use_pe = f->cfg.use_position_encoding;
    aa.n_feat = f->n_feat;
    aa.nb0 = k->nb0;
    aa.nbs = k->nbs;
    aa.emb_dim = f->cfg.appearance_dim;
    aa.ref_flags = f->ref_flags;
    aa.grad = grad;
    aa.csmall_tp = w.csmall;
    aa.g_save = w.gsave;  // null unless a backward follows AND the colour inputs depend on the normal (carve)
    aa.d_save = w.dsave;
    { ProfScope ps_(PS_ASSEMBLE, s); grad_assemble_kernel<<<(unsigned)(NP / 256 + (NP % 256 != 0)), 256, 0, s>>>(aa); }

    ColFwdArgs ca;
    memset(&ca, 0, sizeof(ca));
    fill_col_ptrs(f, packed, &ca.p, kNsCol);
    ca.feat_tp = w.feat;
    ca.csmall_tp = w.csmall;
    for (int l = 0; l < f->nlc; ++l) ca.h_tp[l] = w.h[l];
    ca.rgb = w.rgb;  // kept for the backward's sigmoid derivative; the caller gets a copy
    { ProfScope ps_(PS_COL_FWD, s); k->col_fwd(ca, save, grid, s); }
    SDFHIP_CHECK_HIP(hipMemcpyAsync(rgb, w.rgb, (size_t)NP * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  if (feat != nullptr && mode != SDFHIP_MODE_SDF) {
    const int64_t total = P * f->cfg.geo_feat_dim;
    untp_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(w.feat, k->nbf, f->cfg.geo_feat_dim, P, feat);
  }
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// block = 32 columns x 32 split lanes (a thread sums n_split / 32 partials, the lanes combine through LDS); grid = ceil(stride / 32).
// (One thread per column over all 256 splits was 256 dependent round trips: 0.10 ms for 288 sums.)
__global__ __launch_bounds__(1024) void sdfrow_reduce_kernel(const float* __restrict__ partial, const int n_split, const int stride,
                                                             const int hidden, float* __restrict__ w_row, float* __restrict__ b0) {
  __shared__ float red[32][33];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), r0 = threadIdx.x >> 5;
  float s = 0.0f;
  if (c < stride)
    for (int k = r0; k < n_split; k += 32) s += partial[(size_t)k * stride + c];
  red[r0][threadIdx.x & 31] = s;
  __syncthreads();
  if (r0 == 0 && c < stride) {
#pragma unroll
    for (int r = 1; r < 32; ++r) s += red[r][threadIdx.x & 31];
    if (c < hidden) w_row[c] = s;
    if (c == stride - 32) b0[0] = s;
  }
}

static void flush_wreduce(WreduceBatch& B, hipStream_t s);
static void run_wgrad(const SdfHipField* f, const FieldWs& w, const WgradArgs& base, int rowmap, int colmap, int64_t w_off, int ld,
                      float scale, int64_t b_off, float* theta_bar, hipStream_t s) {
  WgradArgs a = base;
  a.tiles_per_split = (int)((a.n_tiles + w.n_split - 1) / w.n_split);
  // a region of its own and a deferred reduction (flush_wreduce), or the shared buffer and the reduction right behind the GEMM
  WreduceBatch* B = w.batch;
  const int64_t need = (int64_t)w.n_split * a.nba * 32 * a.nbb * 32, bneed = (int64_t)w.n_split * a.nba * 32;
  if (B != nullptr && (B->args.n >= kWreduceBatchMax || B->used + need > B->cap || B->bused + bneed > B->bcap))
    flush_wreduce(*B, s);  // no room left: reduce what is pending (in stream order: before this GEMM reuses the regions) and start over
  const bool defer = B != nullptr && need <= B->cap && bneed <= B->bcap;
  float* const part = defer ? w.partial + B->used : w.partial;
  float* const bpart = defer ? w.bpartial + B->bused : w.bpartial;
  a.partial = part;
  a.bpartial = b_off >= 0 ? bpart : nullptr;
  {
    ProfScope ps_(PS_WGRAD, s);
    a.n_split = w.n_split;
    a.quad = 0;
    if (a.nba == 16 && a.nbb == 16 && w.n_split % 8 == 0) {
      // hidden 512: the four 256 x 256 macro tiles in one launch, interleaved so that they share their operand reads in L2 (WgradArgs::quad)
      a.quad = 1;
      WgradKernelFn fn = wgrad8_pick(2, 4);
      (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, kW8LdsBytes);
      hipLaunchKernelGGL(fn, dim3(4u * (unsigned)w.n_split), dim3(512), kW8LdsBytes, s, a);
    } else {
      for (int ob = 0; ob < a.nba; ob += 8)
        for (int ib = 0; ib < a.nbb; ib += 8) {
          a.ob_base = ob;
          a.ib_base = ib;
          const int rows = std::min(8, a.nba - ob), cols = std::min(8, a.nbb - ib);
          // weight-gradient products: split-bf16 (3 bf16 MFMAs per product, fp32 accumulate) on the 8-wave double-buffered kernel
          WgradKernelFn fn = wgrad8_pick((rows + 3) / 4, (cols + 1) / 2);
          (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, kW8LdsBytes);
          hipLaunchKernelGGL(fn, dim3((unsigned)w.n_split), dim3(512), kW8LdsBytes, s, a);
        }
    }
  }
  WreduceArgs r;
  memset(&r, 0, sizeof(r));
  r.partial = part;
  r.bpartial = a.bpartial;
  r.n_split = w.n_split;
  r.rows = a.nba * 32;
  r.cols = a.nbb * 32;
  r.rowmap = f->d_maps + rowmap;
  r.colmap = f->d_maps + colmap;
  r.theta_bar = theta_bar;
  r.w_off = w_off;
  r.ld = ld;
  r.scale = scale;
  r.b_off = b_off;
  r.accumulate = 0;
  if (defer) {
    WreduceBatchArgs& ba = B->args;
    if (ba.n == 0) ba.first_block[0] = 0;
    ba.r[ba.n] = r;
    ba.first_block[ba.n + 1] = ba.first_block[ba.n] + wreduce_blocks(r);
    ++ba.n;
    B->used += need;
    B->bused += bneed;
    return;
  }
  { ProfScope ps_(PS_WREDUCE, s); wreduce_kernel<<<(unsigned)wreduce_blocks(r), 64 * kWrG, 0, s>>>(r); }
}
// the deferred reductions of one backward call, one launch
static void flush_wreduce(WreduceBatch& B, hipStream_t s) {
  if (B.args.n == 0) return;
  ProfScope ps_(PS_WREDUCE, s);
  wreduce_batch_kernel<<<(unsigned)B.args.first_block[B.args.n], 64 * kWrG, 0, s>>>(B.args);
  B.args.n = 0;
  B.used = B.bused = 0;
}
static void begin_wreduce_batch(const SdfHipField* f, FieldWs& w, WreduceBatch& B) {
  w.batch = nullptr;
  if (!f->batch_wreduce || w.partial == nullptr) return;
  B.args.n = 0;
  B.used = B.bused = 0;
  B.cap = (int64_t)w.n_split * f->total_partial_elems;
  B.bcap = (int64_t)w.n_split * f->total_partial_rows;
  w.batch = &B;
}

static TpOperand seg1(const float* p, int nb) {
  TpOperand o;
  memset(&o, 0, sizeof(o));
  o.ptr[0] = p;
  o.nb[0] = nb;
  return o;
}
static TpOperand seg2(const float* p0, int nb0, const float* p1, int nb1) {
  TpOperand o;
  memset(&o, 0, sizeof(o));
  o.ptr[0] = p0;
  o.nb[0] = nb0;
  o.ptr[1] = p1;
  o.nb[1] = nb1;
  return o;
}

// Weight gradients of the geometry network: split-K GEMMs over the points.  tangent = the second-order pair (R_l^T Qb_l) rides along.
// n_tiles_feat: tiles whose rows carry a feature cotangent (default: all) - the output layer's feature-row GEMM runs over those only.
static void run_geo_wgrads(const SdfHipField* f, const FieldWs& w, const bool tangent, const int64_t n_tiles, float* theta_bar,
                           hipStream_t s, int64_t n_tiles_feat = -1) {
  if (n_tiles_feat < 0) n_tiles_feat = n_tiles;
  const FieldKernels* k = f->k;
  for (int l = 0; l < f->nl; ++l) {
    const LinearInfo& li = f->lin[l];
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.n_pairs = tangent ? 2 : 1;
    a.nba = f->nbo_geo(l);
    a.nbb = f->kb_geo(l);
    a.n_tiles = n_tiles;
    a.A[0] = seg1(w.zb[l], a.nba);
    a.A[1] = seg1(w.r[l], a.nba);
    if (l == 0) {
      a.B[0] = seg1(w.in0, k->nb0);
      a.B[1] = seg1(w.ebar, k->nb0);
    } else if (l == f->skip) {
      a.B[0] = seg2(w.u[l - 1], f->nb3, w.in0, k->nb0);
      a.B[1] = seg1(w.qb[l], a.nbb);
    } else {
      a.B[0] = seg1(w.u[l - 1], a.nbb);
      a.B[1] = seg1(w.qb[l], a.nbb);
    }
    run_wgrad(f, w, a, f->g_rowmap[l], f->g_colmap[l], li.w_off, li.in_dim, f->g_scale[l], li.b_off, theta_bar, s);
  }
  {
    // output layer: feature rows through the GEMM, sdf row through its own reduction
    const LinearInfo& li = f->lin[f->nl];
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.n_pairs = 1;
    a.nba = k->nbf;
    a.nbb = k->nbh;
    a.n_tiles = n_tiles_feat;
    a.A[0] = seg1(w.featbar, k->nbf);
    a.B[0] = seg1(w.u[f->nl - 1], k->nbh);
    run_wgrad(f, w, a, f->g_rowmap[f->nl], f->g_colmap[f->nl], li.w_off, li.in_dim, 1.0f, li.b_off, theta_bar, s);
    const int tps = (int)((n_tiles + w.n_split - 1) / w.n_split);
    { ProfScope ps_(PS_WGRAD, s); k->sdfrow(w.u[f->nl - 1], tangent ? w.qb[f->nl] : nullptr, w.sdfbar, n_tiles, tps, w.sdfrow_partial, (unsigned)w.n_split, s); }
    const int stride = k->nbh * 32 + 32;
    sdfrow_reduce_kernel<<<(stride + 31) / 32, 1024, 0, s>>>(w.sdfrow_partial, w.n_split, stride, f->cfg.hidden_dim, theta_bar + li.w_off,
                                                              theta_bar + li.b_off);
  }
}

// Weight gradients of the colour network (first order only).
static void run_col_wgrads(const SdfHipField* f, const FieldWs& w, const int64_t n_tiles, float* theta_bar, hipStream_t s) {
  const FieldKernels* k = f->k;
  for (int l = 0; l < f->nlc; ++l) {
    const LinearInfo& li = f->lin[f->n_geo + l];
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.n_pairs = 1;
    a.nba = k->nbc;
    a.nbb = f->kb_col(l);
    a.n_tiles = n_tiles;
    a.A[0] = seg1(w.d[l], k->nbc);
    a.B[0] = l == 0 ? seg2(w.feat, k->nbf, w.csmall, k->nbs) : seg1(w.h[l - 1], k->nbc);
    run_wgrad(f, w, a, f->c_rowmap[l], f->c_colmap[l], li.w_off, li.in_dim, 1.0f, li.b_off, theta_bar, s);
  }
  {
    const LinearInfo& li = f->lin[f->n_geo + f->nlc];
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.n_pairs = 1;
    a.nba = 1;
    a.nbb = k->nbc;
    a.n_tiles = n_tiles;
    a.A[0] = seg1(w.dout, 1);
    a.B[0] = seg1(w.h[f->nlc - 1], k->nbc);
    run_wgrad(f, w, a, f->c_rowmap[f->nlc], f->c_colmap[f->nlc], li.w_off, li.in_dim, 1.0f, li.b_off, theta_bar, s);
  }
}

// ------------------------------------------------------------------------------------------------ differentiable geometry network
// forward_geonetwork (sdf_field.py:380-410) as a first-order differentiable operator on explicit positions: what the reference
// differentiates through in the sparse-SfM loss (base_surface_model.py:463), and the building block of the numerical-gradient
// path (sdf_field.py:433-453: six more evaluations of the same network).
static void carve_geo(const SdfHipField* f, int64_t n_points, void* base, FieldWs* w) {
  const FieldKernels* k = f->k;
  const int64_t np = sdfhip_padded_points(n_points);
  size_t off = 0;
  auto take = [&](int64_t floats) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += ((size_t)floats * sizeof(float) + 255) / 256 * 256;
    return p;
  };
  memset(w, 0, sizeof(*w));
  w->x = take(np * 3);
  w->in0 = take(np * k->nb0 * 32);
  w->feat = take(np * k->nbf * 32);
  for (int l = 0; l < f->nl; ++l) w->u[l] = take(np * f->nbo_geo(l) * 32);
  w->sdfbar = take(np);
  for (int l = 0; l < f->nl; ++l) w->zb[l] = take(np * f->nbo_geo(l) * 32);
  w->in0bar = take(np * k->nb0 * 32);
  w->featbar = take(np * k->nbf * 32);
  const int64_t n_tiles = np / 32;
  w->n_split = (int)std::min<int64_t>(256, n_tiles);
  w->partial = take((int64_t)w->n_split * partial_floats_per_split(f));
  w->bpartial = take((int64_t)w->n_split * bpartial_floats_per_split(f));
  w->sdfrow_partial = sdfrow_region(f, w->partial, w->n_split);
  w->bytes = off;
}

extern "C" int64_t sdfhip_geo_workspace_size(const SdfHipField* f, int64_t n_points) {
  FieldWs w;
  carve_geo(f, n_points, nullptr, &w);
  return (int64_t)w.bytes;
}

// natural [n_rows][n_feat] (or null: zeros) -> tile-packed [T][nb], rows >= n_rows and features >= n_feat zero
__global__ void totp_kernel(const float* __restrict__ in, const int nb, const int n_feat, const int64_t n_rows, const int64_t n_padded,
                            float* __restrict__ tp) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int width = nb * 32;
  if (idx >= n_padded * width) return;
  const int64_t p = idx / width;
  const int c = (int)(idx % width);
  tp[tp_index(p, c, nb)] = (in != nullptr && p < n_rows && c < n_feat) ? in[p * n_feat + c] : 0.0f;
}
__global__ void pad_copy_kernel(const float* __restrict__ in, const int64_t n, const int64_t n_padded, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_padded) out[i] = (in != nullptr && i < n) ? in[i] : 0.0f;
}

// shared by the explicit-position entry (dirs == null: `origins` holds [P,3] positions, used as given) and the ray entry (positions from
// the rays' frustums: start points, or mid points when `ends` is given; the field's scene contraction applied)
static int geo_forward_impl(const SdfHipField* f, const float* packed, const float* table, const float* level_mask, const float* origins,
                            const float* dirs, const float* starts, const float* ends, int32_t n_samples, int64_t n_points,
                            int64_t n_feat_points, void* workspace, float* sdf, float* feat, float* x_out, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(f && packed && table && level_mask && origins && workspace && sdf, "geo_forward: null argument");
  SDFHIP_REQUIRE(n_feat_points >= 0 && n_feat_points <= n_points, "geo_forward: n_feat_points out of range");
  if (n_points == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const FieldKernels* k = f->k;
  const int64_t P = n_points, NP = sdfhip_padded_points(P);
  FieldWs w;
  carve_geo(f, P, workspace, &w);
  EncodeArgs ea;
  memset(&ea, 0, sizeof(ea));
  ea.grid = f->grid;
  ea.origins = origins;  // explicit positions are used as given (forward_geonetwork does not contract)
  ea.dirs = dirs;
  ea.starts = starts;
  ea.ends = ends;
  ea.contract = dirs != nullptr ? f->cfg.contract : 0;
  ea.n_points = P;
  ea.n_padded = NP;
  ea.S = dirs != nullptr ? n_samples : 1;
  ea.pe_degree = f->pe_code;
  ea.use_pe = f->cfg.use_position_encoding;
  ea.nb0 = k->nb0;
  ea.table = table;
  ea.mask = level_mask;
  ea.x_out = w.x;
  ea.in0_tp = w.in0;
  {
    ProfScope ps_(PS_ENCODE, s);
    const unsigned gx = (unsigned)(NP / 256 + (NP % 256 != 0));
    if (f->grid.n_features == 8) geo_encode8_kernel<<<dim3(gx, f->grid.n_levels + 1), 256, 0, s>>>(ea);
    else geo_encode_kernel<<<dim3(gx, f->grid.n_levels * (f->grid.n_features / 2) + 1), 256, 0, s>>>(ea);
  }
  GeoFwdArgs ga;
  memset(&ga, 0, sizeof(ga));
  fill_geo_ptrs(f, packed, &ga.p, kNsFwd);
  ga.in0_tp = w.in0;
  for (int l = 0; l < f->nl; ++l) ga.u_tp[l] = w.u[l];
  ga.feat_tp = w.feat;
  ga.sdf = sdf;
  { ProfScope ps_(PS_GEO_FWD, s); k->geo_fwd(3, ga, (unsigned)(NP / 128), s); }
  if (feat != nullptr && n_feat_points > 0) {
    const int64_t total = n_feat_points * f->cfg.geo_feat_dim;
    untp_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(w.feat, k->nbf, f->cfg.geo_feat_dim, n_feat_points, feat);
  }
  if (x_out != nullptr) SDFHIP_CHECK_HIP(hipMemcpyAsync(x_out, w.x, (size_t)P * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_geo_forward_n(const SdfHipField* f, const float* packed, const float* table, const float* level_mask,
                                    const float* positions, int64_t n_points, int64_t n_feat_points, void* workspace, float* sdf,
                                    float* feat, sdfhip_stream_t stream) {
  return geo_forward_impl(f, packed, table, level_mask, positions, nullptr, nullptr, nullptr, 1, n_points, n_feat_points, workspace, sdf, feat,
                          nullptr, stream);
}
extern "C" int sdfhip_geo_forward_rays(const SdfHipField* f, const float* packed, const float* table, const float* level_mask,
                                       const float* origins, const float* dirs, const float* starts, const float* ends, int64_t n_rays,
                                       int32_t n_samples, void* workspace, float* sdf, float* feat, float* x_out, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(dirs && starts && n_rays >= 0 && n_samples > 0, "geo_forward_rays: bad argument");
  const int64_t P = n_rays * n_samples;
  return geo_forward_impl(f, packed, table, level_mask, origins, dirs, starts, ends, n_samples, P, P, workspace, sdf, feat, x_out, stream);
}
extern "C" int sdfhip_geo_forward(const SdfHipField* f, const float* packed, const float* table, const float* level_mask,
                                  const float* positions, int64_t n_points, void* workspace, float* sdf, float* feat,
                                  sdfhip_stream_t stream) {
  return sdfhip_geo_forward_n(f, packed, table, level_mask, positions, n_points, n_points, workspace, sdf, feat, stream);
}

extern "C" int sdfhip_geo_backward_n(const SdfHipField* f, const float* packed, const float* level_mask, int64_t n_points, void* workspace,
                                     int64_t n_feat_points, const float* sdf_bar, const float* feat_bar, float* theta_bar, float* table_bar,
                                   sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(f && packed && level_mask && workspace && theta_bar && table_bar, "geo_backward: null argument");
  SDFHIP_REQUIRE(n_feat_points >= 0 && n_feat_points <= n_points, "geo_backward: n_feat_points out of range");
  if (n_points == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const FieldKernels* k = f->k;
  const int64_t P = n_points, NP = sdfhip_padded_points(P);
  FieldWs w;
  carve_geo(f, P, workspace, &w);
  const unsigned pg = (unsigned)((NP + 255) / 256);
  pad_copy_kernel<<<pg, 256, 0, s>>>(sdf_bar, P, NP, w.sdfbar);
  const int64_t fw = NP * k->nbf * 32;
  // rows >= n_feat_points (points whose feature the caller never took: the six taps of the numerical gradient) get zeros: the
  // tile-packed layout is tile-major, so everything behind the last tile with a live row is ONE contiguous memset
  {
    const int64_t n_conv = std::min<int64_t>(NP, (n_feat_points + 31) / 32 * 32);
    const int64_t cw = n_conv * k->nbf * 32;
    if (cw > 0) totp_kernel<<<(unsigned)((cw + 255) / 256), 256, 0, s>>>(feat_bar, k->nbf, f->cfg.geo_feat_dim, n_feat_points, n_conv, w.featbar);
    if (fw > cw) SDFHIP_CHECK_HIP(hipMemsetAsync(w.featbar + cw, 0, (size_t)(fw - cw) * sizeof(float), s));
  }

  GeoBwdArgs gb;
  memset(&gb, 0, sizeof(gb));
  fill_geo_ptrs(f, packed, &gb.p, kNsGrad);
  gb.featbar_tp = w.featbar;
  gb.sdfbar = w.sdfbar;
  for (int l = 0; l < f->nl; ++l) {
    gb.u_tp[l] = w.u[l];
    gb.zb_tp[l] = w.zb[l];
  }
  gb.in0bar_tp = w.in0bar;
  { ProfScope ps_(PS_GEO_BWD, s); k->geo_bwd1(gb, (unsigned)(NP / 128), s); }

  GridBwdArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.grid = f->grid;
  ga.x = w.x;
  ga.in0bar_tp = w.in0bar;
  ga.mask = level_mask;
  ga.n_points = P;
  ga.pe_degree = f->pe_code;
  ga.nb0 = k->nb0;
  ga.tablebar = table_bar;
  if (f->grid.n_levels > 0) {  // NeRFField has no grid
    ProfScope ps_(PS_GRID_BWD, s);
    if (f->grid.n_features == 8) grid_bwd8_kernel<<<dim3((unsigned)((P + 255) / 256), f->grid.n_levels), 256, 0, s>>>(ga);
    else grid_bwd_kernel<<<dim3((unsigned)((P + 255) / 256), f->grid.n_levels * (f->grid.n_features / 2)), 256, 0, s>>>(ga);
  }

  run_geo_wgrads(f, w, false, NP / 32, theta_bar, s);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int sdfhip_geo_backward(const SdfHipField* f, const float* packed, const float* level_mask, int64_t n_points, void* workspace,
                                   const float* sdf_bar, const float* feat_bar, float* theta_bar, float* table_bar,
                                   sdfhip_stream_t stream) {
  return sdfhip_geo_backward_n(f, packed, level_mask, n_points, workspace, n_points, sdf_bar, feat_bar, theta_bar, table_bar, stream);
}

// ------------------------------------------------------------------------------------------------ colour network as its own operator
// SDFField.get_colors (sdf_field.py:532-612) with every input supplied by the caller: the numerical-gradient path feeds it the
// finite-difference normal instead of the analytic one (sdf_field.py:639-644, 655).
static void carve_col(const SdfHipField* f, int64_t n_points, void* base, FieldWs* w) {
  const FieldKernels* k = f->k;
  const int64_t np = sdfhip_padded_points(n_points);
  size_t off = 0;
  auto take = [&](int64_t floats) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += ((size_t)floats * sizeof(float) + 255) / 256 * 256;
    return p;
  };
  memset(w, 0, sizeof(*w));
  w->feat = take(np * k->nbf * 32);
  w->csmall = take(np * k->nbs * 32);
  for (int l = 0; l < f->nlc; ++l) w->h[l] = take(np * k->nbc * 32);
  w->rgb = take(np * 3);
  for (int l = 0; l < f->nlc; ++l) w->d[l] = take(np * k->nbc * 32);
  w->dout = take(np * 32);
  w->featbar = take(np * k->nbf * 32);
  w->csmallbar = take(np * k->nbs * 32);
  const int64_t n_tiles = np / 32;
  w->n_split = (int)std::min<int64_t>(256, n_tiles);
  w->partial = take((int64_t)w->n_split * partial_floats_per_split(f));
  w->bpartial = take((int64_t)w->n_split * bpartial_floats_per_split(f));
  w->sdfrow_partial = sdfrow_region(f, w->partial, w->n_split);
  w->bytes = off;
}

extern "C" int64_t sdfhip_color_workspace_size(const SdfHipField* f, int64_t n_points) {
  FieldWs w;
  carve_col(f, n_points, nullptr, &w);
  return (int64_t)w.bytes;
}

extern "C" int sdfhip_color_forward(const SdfHipField* f, const float* packed, const float* feat, const float* x, const float* dirs,
                                    const float* grad, const float* emb, int64_t n_rays, int32_t n_samples, void* workspace,
                                    float* rgb, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(f && packed && feat && x && dirs && grad && workspace && rgb, "color_forward: null argument");
  SDFHIP_REQUIRE(n_samples >= 1 && n_rays >= 0, "color_forward: bad shape");
  if (n_rays == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const FieldKernels* k = f->k;
  const int64_t P = n_rays * n_samples, NP = sdfhip_padded_points(P);
  FieldWs w;
  carve_col(f, P, workspace, &w);
  const int64_t fw = NP * k->nbf * 32;
  totp_kernel<<<(unsigned)((fw + 255) / 256), 256, 0, s>>>(feat, k->nbf, f->cfg.geo_feat_dim, P, NP, w.feat);
  AssembleArgs aa;
  memset(&aa, 0, sizeof(aa));
  aa.x = x;
  aa.dirs = dirs;
  aa.emb = emb;
  aa.grad_in = grad;
  aa.n_points = P;
  aa.n_padded = NP;
  aa.S = n_samples;
  aa.pe_degree = f->pe_code;
  aa.use_pe = f->cfg.use_position_encoding;
  aa.n_feat = f->n_feat;
  aa.nb0 = k->nb0;
  aa.nbs = k->nbs;
  aa.emb_dim = f->cfg.appearance_dim;
  aa.ref_flags = f->ref_flags;
  aa.csmall_tp = w.csmall;
  { ProfScope ps_(PS_ASSEMBLE, s); grad_assemble_kernel<<<(unsigned)(NP / 256 + (NP % 256 != 0)), 256, 0, s>>>(aa); }
  ColFwdArgs ca;
  memset(&ca, 0, sizeof(ca));
  fill_col_ptrs(f, packed, &ca.p, kNsCol);
  ca.feat_tp = w.feat;
  ca.csmall_tp = w.csmall;
  for (int l = 0; l < f->nlc; ++l) ca.h_tp[l] = w.h[l];
  ca.rgb = w.rgb;
  { ProfScope ps_(PS_COL_FWD, s); k->col_fwd(ca, 1, (unsigned)(NP / 128), s); }
  SDFHIP_CHECK_HIP(hipMemcpyAsync(rgb, w.rgb, (size_t)NP * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// d L / d (normal input) [P,3] and d L / d (appearance embedding) [N, emb_dim] out of the colour backward's small-input block
__global__ void color_unpack_kernel(const float* __restrict__ csmallbar_tp, const int nbs, const int64_t n_points, const int g_col,
                                    float* __restrict__ grad_bar) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_points) return;
#pragma unroll
  for (int d = 0; d < 3; ++d) grad_bar[p * 3 + d] = g_col >= 0 ? csmallbar_tp[tp_index(p, g_col + d, nbs)] : 0.0f;
}
// d L / d (per-ray embedding) [N, emb_dim] += sum over the ray's S consecutive points: one thread per (ray, slot), no atomics (the
// caller's buffer is zero or holds an earlier contribution)
__global__ void color_emb_reduce_kernel(const float* __restrict__ csmallbar_tp, const int nbs, const int64_t n_rays, const int S,
                                        const int emb_dim, const int emb_col, float* __restrict__ emb_bar) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rays * emb_dim) return;
  const int64_t ray = idx / emb_dim;
  const int j = (int)(idx % emb_dim);
  float s = 0.0f;
  for (int i = 0; i < S; ++i) s += csmallbar_tp[tp_index(ray * S + i, emb_col + j, nbs)];
  emb_bar[idx] += s;
}

extern "C" int sdfhip_color_backward(const SdfHipField* f, const float* packed, int64_t n_rays, int32_t n_samples, void* workspace,
                                     const float* rgb_bar, float* theta_bar, float* feat_bar, float* grad_bar, float* emb_bar,
                                     sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(f && packed && workspace && rgb_bar && theta_bar, "color_backward: null argument");
  SDFHIP_REQUIRE((f->ref_flags & (kRefReflect | kRefNdotV)) == 0,
                 "color_backward: use_reflections / use_n_dot_v send the normal's cotangent through the reflected direction - built in "
                 "sdfhip_field_backward (analytic normals) only");
  if (n_rays == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const FieldKernels* k = f->k;
  const int64_t P = n_rays * n_samples, NP = sdfhip_padded_points(P);
  FieldWs w;
  carve_col(f, P, workspace, &w);
  ColBwdArgs cb;
  memset(&cb, 0, sizeof(cb));
  fill_col_ptrs(f, packed, &cb.p, kNsGrad);
  cb.rgb = w.rgb;
  cb.rgbbar = rgb_bar;
  cb.n_points = P;
  for (int l = 0; l < f->nlc; ++l) {
    cb.h_tp[l] = w.h[l];
    cb.d_tp[l] = w.d[l];
  }
  cb.dout_tp = w.dout;
  cb.featbar_tp = w.featbar;
  cb.csmallbar_tp = w.csmallbar;
  { ProfScope ps_(PS_COL_BWD, s); k->col_bwd(cb, (unsigned)(NP / 128), s); }
  if (feat_bar != nullptr) {
    const int64_t total = P * f->cfg.geo_feat_dim;
    untp_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(w.featbar, k->nbf, f->cfg.geo_feat_dim, P, feat_bar);
  }
  const CsmallLayout CL = csmall_layout(f->ref_flags, f->cfg.appearance_dim);
  if (grad_bar != nullptr) color_unpack_kernel<<<(unsigned)((P + 255) / 256), 256, 0, s>>>(w.csmallbar, k->nbs, P, CL.g, grad_bar);
  if (emb_bar != nullptr && f->cfg.appearance_dim > 0) {
    const int64_t ne = n_rays * f->cfg.appearance_dim;
    color_emb_reduce_kernel<<<(unsigned)((ne + 255) / 256), 256, 0, s>>>(w.csmallbar, k->nbs, n_rays, n_samples, f->cfg.appearance_dim, CL.emb,
                                                                          emb_bar);
  }
  run_col_wgrads(f, w, NP / 32, theta_bar, s);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// featbar_tp += feat_bar (natural [P, GF] -> tile-packed), rows >= n_points untouched
__global__ void tp_add_kernel(const float* __restrict__ nat, const int nb, const int n_feat, const int64_t n_rows, float* __restrict__ tp) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * n_feat) return;
  const int64_t p = idx / n_feat;
  const int c = (int)(idx % n_feat);
  tp[tp_index(p, c, nb)] += nat[idx];
}

extern "C" int sdfhip_field_backward(const SdfHipField* f, const float* packed, const float* table, const float* level_mask,
                                     int64_t n_rays, int32_t n_samples, void* workspace, const float* sdf_bar, const float* grad_bar,
                                     const float* rgb_bar, float* theta_bar, float* table_bar, float* emb_bar,
                                     sdfhip_stream_t stream) {
  return sdfhip_field_backward_feat(f, packed, table, level_mask, n_rays, n_samples, workspace, sdf_bar, grad_bar, rgb_bar, nullptr, theta_bar,
                                    table_bar, emb_bar, stream);
}

extern "C" int sdfhip_field_backward_feat(const SdfHipField* f, const float* packed, const float* table, const float* level_mask,
                                          int64_t n_rays, int32_t n_samples, void* workspace, const float* sdf_bar, const float* grad_bar,
                                          const float* rgb_bar, const float* feat_bar, float* theta_bar, float* table_bar, float* emb_bar,
                                          sdfhip_stream_t stream) {
  (void)table;
  SDFHIP_REQUIRE(f && packed && level_mask && workspace && theta_bar && table_bar, "field_backward: null argument");
  SDFHIP_REQUIRE(f->k->geo_bwd != nullptr, "field_backward: no second-order kernels for this (ReLU) field");
  if (n_rays == 0) {  // no points: theta_bar is OVERWRITTEN by this call (table_bar / emb_bar are accumulated into) - the sum over no points is 0
    SDFHIP_CHECK_HIP(hipMemsetAsync(theta_bar, 0, sizeof(float) * (size_t)f->theta_size, (hipStream_t)stream));
    return 0;
  }
  hipStream_t s = (hipStream_t)stream;
  const FieldKernels* k = f->k;
  const int64_t P = n_rays * n_samples, NP = sdfhip_padded_points(P);
  FieldWs w;
  carve(f, P, 1, workspace, &w);
  const unsigned grid = (unsigned)(NP / 128);
  const unsigned pgrid = (unsigned)(NP / 256 + (NP % 256 != 0));

  // 1. colour network backward
  ColBwdArgs cb;
  memset(&cb, 0, sizeof(cb));
  fill_col_ptrs(f, packed, &cb.p, kNsGrad);
  cb.rgb = w.rgb;
  cb.rgbbar = rgb_bar;
  cb.n_points = rgb_bar != nullptr ? P : 0;
  for (int l = 0; l < f->nlc; ++l) {
    cb.h_tp[l] = w.h[l];
    cb.d_tp[l] = w.d[l];
  }
  cb.dout_tp = w.dout;
  cb.featbar_tp = w.featbar;
  cb.csmallbar_tp = w.csmallbar;
  { ProfScope ps_(PS_COL_BWD, s); k->col_bwd(cb, grid, s); }
  if (feat_bar != nullptr) {  // a consumer of the geometry feature outside the colour network (the ref-nerf diffuse / tint heads)
    const int64_t total = P * f->cfg.geo_feat_dim;
    tp_add_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(feat_bar, k->nbf, f->cfg.geo_feat_dim, P, w.featbar);
  }

  // 2. total d L / d grad, tangent seed, padded sdfbar
  BwdPrepArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.gradbar = grad_bar;
  pa.sdfbar_in = sdf_bar;
  pa.csmallbar_tp = w.csmallbar;
  pa.x = w.x;
  pa.dydp = w.dydp;
  pa.mask = level_mask;
  pa.n_points = P;
  pa.n_padded = NP;
  pa.pe_degree = f->pe_code;
  pa.use_pe = f->cfg.use_position_encoding;
  pa.n_feat = f->n_feat;
  pa.nb0 = k->nb0;
  pa.nbs = k->nbs;
  pa.emb_dim = f->cfg.appearance_dim;
  pa.S = n_samples;
  pa.ref_flags = f->ref_flags;
  pa.g_pt = w.gsave;
  pa.d_pt = w.dsave;
  pa.gtot = w.gtot;
  pa.ebar_tp = w.ebar;
  pa.sdfbar = w.sdfbar;
  pa.embbar = emb_bar;
  { ProfScope ps_(PS_BWD_PREP, s); bwd_prep_kernel<<<pgrid, 256, 0, s>>>(pa); }

  // 3. geometry network: tangent pass + data backward
  GeoBwdArgs gb;
  memset(&gb, 0, sizeof(gb));
  fill_geo_ptrs(f, packed, &gb.p, kNsGrad);
  gb.ebar_tp = w.ebar;
  gb.featbar_tp = w.featbar;
  gb.sdfbar = w.sdfbar;
  for (int l = 0; l < f->nl; ++l) {
    gb.u_tp[l] = w.u[l];
    gb.r_tp[l] = w.r[l];
    gb.zb_tp[l] = w.zb[l];
  }
  gb.qb_tp[0] = w.ebar;  // the tangent entering layer 0 IS the seed: the in0 gemm of the tangent pass rewrites it with itself
  for (int l = 1; l <= f->nl; ++l) gb.qb_tp[l] = w.qb[l];
  gb.in0bar_tp = w.in0bar;
  { ProfScope ps_(PS_GEO_BWD, s); k->geo_bwd(gb, grid, s); }

  // 4. hash table gradient (first-order through the features + second-order through d feature / d x)
  GridBwdArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.grid = f->grid;
  ga.x = w.x;
  ga.in0bar_tp = w.in0bar;
  ga.e_tp = w.e;
  ga.gtot = w.gtot;
  ga.mask = level_mask;
  ga.n_points = P;
  ga.pe_degree = f->pe_code;
  ga.nb0 = k->nb0;
  ga.tablebar = table_bar;
  // forked: nothing below reads table_bar, and the scatter is bound by memory-side atomics, not by CUs
  const bool forked = g_side.ready();
  hipStream_t gs = s;
  if (forked) {
    SDFHIP_CHECK_HIP(hipEventRecord(g_side.fork, s));
    SDFHIP_CHECK_HIP(hipStreamWaitEvent(g_side.stream, g_side.fork, 0));
    gs = g_side.stream;
  }
  {
    ProfScope ps_(PS_GRID_BWD, gs);
    if (f->grid.n_features == 8) grid_bwd8_kernel<<<dim3((unsigned)((P + 255) / 256), f->grid.n_levels), 256, 0, gs>>>(ga);
    else grid_bwd_kernel<<<dim3((unsigned)((P + 255) / 256), f->grid.n_levels * (f->grid.n_features / 2)), 256, 0, gs>>>(ga);
  }
  if (forked) SDFHIP_CHECK_HIP(hipEventRecord(g_side.join, gs));
  notify_table_grad(f, table_bar, gs);  // the table's gradient is complete on `gs` from here on: a host may start its exchange behind it

  // 5. weight gradients: split-K GEMMs over points
  const int64_t n_tiles = NP / 32;
  WreduceBatch wb;
  begin_wreduce_batch(f, w, wb);
  run_geo_wgrads(f, w, true, n_tiles, theta_bar, s);
  run_col_wgrads(f, w, n_tiles, theta_bar, s);
  flush_wreduce(wb, s);
  if (forked) SDFHIP_CHECK_HIP(hipStreamWaitEvent(s, g_side.join, 0));
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ small per-ray operators of the "grid" background
// tiny-cuda-nn's SphericalHarmonics encoding of degree 4 (16 real harmonics; oracle/sdf_path.py::sh_degree4 is the statement it is held
// to) of the view direction as fields/nerfacto_field.py:128-134,283-285 feeds it - get_normalized_directions(d) = (d + 1) / 2, mapped back
// to [-1, 1] inside - next to the per-ray appearance embedding: the colour kernel's per-ray slots [SH(16) | emb] in one launch.
__global__ void sh4_embed_kernel(const float* __restrict__ dirs, const float* __restrict__ emb, const int64_t n, const int emb_dim,
                                 float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) v[d] = __fsub_rn(__fmul_rn(__fdiv_rn(__fadd_rn(dirs[i * 3 + d], 1.0f), 2.0f), 2.0f), 1.0f);
  const float x = v[0], y = v[1], z = v[2];
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  float* o = out + i * (16 + emb_dim);
  o[0] = 0.28209479177387814f;
  o[1] = -0.48860251190291987f * y;
  o[2] = 0.48860251190291987f * z;
  o[3] = -0.48860251190291987f * x;
  o[4] = 1.0925484305920792f * xy;
  o[5] = -1.0925484305920792f * yz;
  o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
  o[7] = -1.0925484305920792f * xz;
  o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
  o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
  o[10] = 2.8906114426405538f * xy * z;
  o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
  o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
  o[14] = 1.4453057213202769f * z * (x2 - y2);
  o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
  for (int j = 0; j < emb_dim; ++j) o[16 + j] = emb != nullptr ? emb[i * emb_dim + j] : 0.0f;
}
extern "C" int sdfhip_sh4_embed(const float* dirs, const float* emb, int64_t n_rays, int32_t emb_dim, float* out, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(dirs && out && emb_dim >= 0 && n_rays >= 0, "sh4_embed: bad argument");
  if (n_rays == 0) return 0;
  sh4_embed_kernel<<<(unsigned)((n_rays + 255) / 256), 256, 0, (hipStream_t)stream>>>(dirs, emb, n_rays, emb_dim, out);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// Pinhole rays for a batch of (camera, pixel) draws: the per-batch work of the reference's PixelSampler + RayGenerator (data/utils/
// pixel_samplers.py:47-50 uniform (camera, y, x); model_components/ray_generators.py:49-63 -> cameras.py:462-640 _generate_rays_from_coords
// for a perspective camera without distortion: direction = R [(x + 0.5 - cx) / fx, (y + 0.5 - cy) / fy, 1] with camera-to-world columns
// (x right, y down, z forward), normalised; directions_norm kept).  u [n,3] uniforms in [0,1): camera = floor(u0 C), y = floor(u1 H),
// x = floor(u2 W).  One thread per ray.
struct GenRaysArgs {
  const float* u;        // [n,3]
  const float* centers;  // [C,3]
  const float* rot;      // [C,3,3] row-major camera-to-world
  float fx, fy, cx, cy;
  int32_t n_cams, H, W;
  int64_t n;
  float* origins;  // [n,3]
  float* dirs;     // [n,3] unit
  float* norm;     // [n]   length of the un-normalised direction
  int64_t* cam;    // [n]
};
__global__ void generate_rays_kernel(const GenRaysArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  int c = (int)(a.u[i * 3 + 0] * (float)a.n_cams);
  c = c > a.n_cams - 1 ? a.n_cams - 1 : c;
  const float y = floorf(a.u[i * 3 + 1] * (float)a.H) + 0.5f, x = floorf(a.u[i * 3 + 2] * (float)a.W) + 0.5f;
  const float dc[3] = {(x - a.cx) / a.fx, (y - a.cy) / a.fy, 1.0f};
  const float* R = a.rot + (size_t)c * 9;
  float d[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) d[r] = (R[r * 3 + 0] * dc[0] + R[r * 3 + 1] * dc[1]) + R[r * 3 + 2] * dc[2];
  const float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    a.origins[i * 3 + r] = a.centers[(size_t)c * 3 + r];
    a.dirs[i * 3 + r] = d[r] / len;
  }
  a.norm[i] = len;
  a.cam[i] = c;
}
extern "C" int sdfhip_generate_rays(const float* u, const float* centers, const float* rot, int32_t n_cams, int32_t height, int32_t width,
                                    float fx, float fy, float cx, float cy, int64_t n_rays, float* origins, float* dirs, float* norm,
                                    int64_t* cam, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(u && centers && rot && origins && dirs && norm && cam && n_cams > 0 && height > 0 && width > 0 && n_rays >= 0,
                 "generate_rays: bad argument");
  if (n_rays == 0) return 0;
  GenRaysArgs a{u, centers, rot, fx, fy, cx, cy, n_cams, height, width, n_rays, origins, dirs, norm, cam};
  generate_rays_kernel<<<(unsigned)((n_rays + 255) / 256), 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// Backward of an embedding lookup rows = weight[idx] (field_components/embedding.py; per-camera appearance embeddings, 49 rows x 32):
// out[r] = sum over the n with idx[n] == r of grad[n], one block per table row, a FIXED summation order (torch's
// embedding_backward_feature_kernel: 0.11 ms per call and atomics).  out is overwritten.
__global__ void embedding_backward_kernel(const int64_t* __restrict__ idx, const float* __restrict__ grad, const int64_t n, const int dim,
                                          float* __restrict__ out) {
  const int64_t row = blockIdx.x;
  __shared__ float red[8][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;  // 8 groups of 64 columns
  float s = 0.0f;
  for (int64_t i = g; i < n; i += 8)
    if (idx[i] == row && c < dim) s += grad[i * dim + c];
  red[g][c] = s;
  __syncthreads();
  if (g == 0 && c < dim) {
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][c];
    out[row * dim + c] = s;
  }
}
extern "C" int sdfhip_embedding_backward(const int64_t* idx, const float* grad, int64_t n, int32_t dim, int64_t n_rows, float* out,
                                         sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(idx && grad && out && dim >= 1 && dim <= 64 && n_rows >= 1 && n >= 0, "embedding_backward: bad argument (dim <= 64)");
  embedding_backward_kernel<<<(unsigned)n_rows, 512, 0, (hipStream_t)stream>>>(idx, grad, n, dim, out);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ numerical-gradient field, one operator
// SDFField.get_outputs with use_numerical_gradients (sdf_field.py:629-655; neus-facto-angelo, BASELINE config 5) as ONE forward and ONE
// backward call: the geometry network on the P contracted sample positions and their six taps (7 P points, tap-major: the centre points
// occupy the first tiles), the finite-difference normal, the colour network on it.  Round 3 composed this in Python out of
// sdfhip_geo_forward_n + torch ops + sdfhip_color_forward, which moved the geometry feature out of the kernels' tile-packed layout and
// back (untp / totp: 1.0 ms of a 9.9 ms step), packed the weights twice and cost ~40 small ATen launches each way.  Here the colour
// kernels read the feature tiles the geometry kernel wrote (and hand their cotangent back the same way), the tap points run the
// sdf-row-only kernels (no feature rows computed, stored, back-propagated or multiplied into a weight gradient for 6/7 of the points),
// and the output layer's weight gradient runs over the centre tiles only.
struct NumWs {
  FieldWs g;           // geometry part over n7 = padded(7 P) points: x, in0, feat, u, sdfbar, zb, in0bar, featbar, partial, bpartial
  FieldWs c;           // colour part over nc = padded(P) points: csmall, h, rgb, d, dout, csmallbar; feat / featbar alias g's first tiles
  float* grad_fd;      // [nc][3]
  int64_t nc, n7;
  size_t bytes;
};
constexpr float kHpDelta = 2e-3f;  // numerical-gradient delta (contracted units) below which sdfhip_numfield_forward evaluates with 24-bit products
// training = false: what a forward WITHOUT a backward to follow touches - positions, in0, the geometry feature, the colour network's
// inputs and rgb (no saved activations, no cotangent staging, no split-K partials: ~1 / 6 of the training carve)
static void carve_num(const SdfHipField* f, int64_t P, void* base, NumWs* w, const bool training = true) {
  const FieldKernels* k = f->k;
  const int64_t nc = sdfhip_padded_points(P), n7 = sdfhip_padded_points(7 * P);
  size_t off = 0;
  auto take = [&](int64_t floats) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += ((size_t)floats * sizeof(float) + 255) / 256 * 256;
    return p;
  };
  memset(w, 0, sizeof(*w));
  w->nc = nc;
  w->n7 = n7;
  FieldWs& g = w->g;
  FieldWs& c = w->c;
  g.x = take(n7 * 3);
  g.in0 = take(n7 * k->nb0 * 32);
  g.feat = take(n7 * k->nbf * 32);     // tap tiles stay unwritten when the sdf-row-only forward exists (has_sdf_save)
  if (training || k->layerwise)        // (the layer-at-a-time kernels hand their activations over through these in every mode)
    for (int l = 0; l < f->nl; ++l) g.u[l] = take(n7 * f->nbo_geo(l) * 32);
  if (training) {
    g.sdfbar = take(n7);
    for (int l = 0; l < f->nl; ++l) g.zb[l] = take(n7 * f->nbo_geo(l) * 32);
    g.in0bar = take(n7 * k->nb0 * 32);
    g.featbar = take(n7 * k->nbf * 32);
  }
  c.feat = g.feat;
  c.featbar = g.featbar;
  c.csmall = take(nc * k->nbs * 32);
  if (training)
    for (int l = 0; l < f->nlc; ++l) c.h[l] = take(nc * k->nbc * 32);
  c.rgb = take(nc * 3);
  if (training) {
    for (int l = 0; l < f->nlc; ++l) c.d[l] = take(nc * k->nbc * 32);
    c.dout = take(nc * 32);
    c.csmallbar = take(nc * k->nbs * 32);
  }
  w->grad_fd = take(nc * 3);
  if (training) {
    const int64_t n_tiles = n7 / 32;
    g.n_split = (int)std::min<int64_t>(256, n_tiles);
    g.partial = take((int64_t)g.n_split * partial_floats_per_split(f));
    g.bpartial = take((int64_t)g.n_split * bpartial_floats_per_split(f));
    g.sdfrow_partial = sdfrow_region(f, g.partial, g.n_split);
    c.n_split = (int)std::min<int64_t>(256, nc / 32);
    c.partial = g.partial;  // the weight-gradient GEMMs run one after the other on one stream
    c.bpartial = g.bpartial;
  }
  w->bytes = off;
}
extern "C" int64_t sdfhip_numfield_workspace_size(const SdfHipField* f, int64_t n_points) {
  NumWs w;
  carve_num(f, n_points, nullptr, &w);
  return (int64_t)w.bytes;
}
extern "C" int64_t sdfhip_numfield_inference_workspace_size(const SdfHipField* f, int64_t n_points) {
  NumWs w;
  carve_num(f, n_points, nullptr, &w, false);
  return (int64_t)w.bytes;
}
extern "C" int64_t sdfhip_numfield_sdf_rows(int64_t n_points) { return sdfhip_padded_points(7 * n_points); }

// every tile-indexed pointer of a geometry launch advanced by `tile0` tiles: the launch then covers tiles tile0 .. of the same tensors
static void shift_geo_fwd(const SdfHipField* f, GeoFwdArgs* a, const int64_t tile0) {
  const FieldKernels* k = f->k;
  a->in0_tp += tile0 * k->nb0 * 1024;
  for (int l = 0; l < f->nl; ++l)
    if (a->u_tp[l]) a->u_tp[l] += tile0 * f->nbo_geo(l) * 1024;
  if (a->feat_tp) a->feat_tp += tile0 * k->nbf * 1024;
  a->sdf += tile0 * 32;
}
static void shift_geo_bwd(const SdfHipField* f, GeoBwdArgs* a, const int64_t tile0) {
  const FieldKernels* k = f->k;
  if (a->featbar_tp) a->featbar_tp += tile0 * k->nbf * 1024;
  a->sdfbar += tile0 * 32;
  for (int l = 0; l < f->nl; ++l) {
    a->u_tp[l] += tile0 * f->nbo_geo(l) * 1024;
    a->zb_tp[l] += tile0 * f->nbo_geo(l) * 1024;
  }
  a->in0bar_tp += tile0 * k->nb0 * 1024;
}

extern "C" int sdfhip_numfield_forward(const SdfHipField* f, const float* packed, const float* table, const float* level_mask,
                                       const float* origins, const float* dirs, const float* starts, int64_t n_rays, int32_t n_samples,
                                       const float* emb, float delta, int32_t training, void* workspace, float* sdf7, float* grad,
                                       float* rgb, float* taps, float* x_out, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(f == nullptr || f->ref_flags == 0, "numfield_forward: the ref-nerf colour options are built on the analytic-normal path");
  SDFHIP_REQUIRE(f && packed && table && level_mask && origins && dirs && starts && workspace && sdf7 && grad && rgb,
                 "numfield_forward: null argument");
  SDFHIP_REQUIRE(n_samples >= 1 && n_rays >= 0 && delta > 0.0f, "numfield_forward: bad shape / delta");
  SDFHIP_REQUIRE(!f->k->layerwise, "numfield_forward: the layer-at-a-time (512-wide) kernels are not wired into this operator");
  if (n_rays == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const FieldKernels* k = f->k;
  const int64_t P = n_rays * n_samples;
  NumWs w;
  carve_num(f, P, workspace, &w, training != 0);
  const int64_t NC = w.nc, N7 = w.n7;

  EncodeArgs ea;
  memset(&ea, 0, sizeof(ea));
  ea.grid = f->grid;
  ea.origins = origins;
  ea.dirs = dirs;
  ea.starts = starts;
  ea.n_points = 7 * P;
  ea.n_padded = N7;
  ea.S = n_samples;
  ea.contract = f->cfg.contract;  // get_outputs contracts the sample positions (sdf_field.py:629), the taps are taken in contracted space
  ea.pe_degree = f->pe_code;
  ea.use_pe = f->cfg.use_position_encoding;
  ea.nb0 = k->nb0;
  ea.table = table;
  ea.mask = level_mask;
  ea.x_out = w.g.x;
  ea.in0_tp = w.g.in0;
  ea.tap_points = P;
  ea.tap_delta = delta;
  {
    ProfScope ps_(PS_ENCODE, s);
    const unsigned gx = (unsigned)(N7 / 256 + (N7 % 256 != 0));
    if (f->grid.n_features == 8) geo_encode8_kernel<<<dim3(gx, f->grid.n_levels + 1), 256, 0, s>>>(ea);
    else geo_encode_kernel<<<dim3(gx, f->grid.n_levels * (f->grid.n_features / 2) + 1), 256, 0, s>>>(ea);
  }
  // The numerical normal divides DIFFERENCES of sdf values by 2 delta: an sdf error eps becomes eps / delta in the normal.  At the small
  // deltas of neus-facto-angelo's schedule (down to 2.4e-4 in contracted units) the 22-bit products of the default forward (eps ~ 3e-7)
  // show; below kHpDelta the seven evaluations run with all 24 bits (precision mode 3: six bf16 terms per product - the error class of the
  // fp32 GEMM the reference runs) where the shape has such kernels (FieldKernels::has_hp; SDFHIP_NUMFIELD_HP=0/1 overrides for A/B runs).
  static const int hp_env = [] { const char* e = getenv("SDFHIP_NUMFIELD_HP"); return e ? atoi(e) : -1; }();
  const bool hp = k->has_hp && k->has_sdf_save && (hp_env >= 0 ? hp_env != 0 : delta < kHpDelta);
  GeoFwdArgs ga;
  memset(&ga, 0, sizeof(ga));
  fill_geo_ptrs(f, packed, &ga.p, kNsFwd);
  ga.in0_tp = w.g.in0;
  for (int l = 0; l < f->nl; ++l) ga.u_tp[l] = w.g.u[l];
  ga.feat_tp = w.g.feat;
  ga.sdf = sdf7;
  {
    ProfScope ps_(PS_GEO_FWD, s);
    // centre tiles (and the few tap points that share their last workgroup): sdf + feature; tap tiles: the sdf row alone
    const int save = training != 0;
    if (hp) {
      // feature rows of the centre tiles at the default precision, then the sdf rows of ALL seven evaluations with 24-bit products: the
      // second pass rewrites the centre's sdf and saved activations, so every sdf value and every u_l the backward reads is the 24-bit one
      k->geo_fwd(save ? 3 : 1, ga, (unsigned)(NC / 128), s);
      GeoFwdArgs gh = ga;
      fill_geo_ptrs(f, packed, &gh.p, 3);
      k->geo_fwd((save ? 5 : 2) | kGeoHp, gh, (unsigned)(N7 / 128), s);
    } else if (k->has_sdf_save && N7 > NC) {
      k->geo_fwd(save ? 3 : 1, ga, (unsigned)(NC / 128), s);
      GeoFwdArgs gt = ga;
      shift_geo_fwd(f, &gt, NC / 32);
      k->geo_fwd(save ? 5 : 2, gt, (unsigned)((N7 - NC) / 128), s);
    } else {
      k->geo_fwd(save ? 3 : 1, ga, (unsigned)(N7 / 128), s);
    }
  }
  FdArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.sdf7 = sdf7;
  fa.n_points = P;
  fa.delta = delta;
  fa.grad = grad;
  fa.taps = taps;
  fd_normal_kernel<<<(unsigned)((P + 255) / 256), 256, 0, s>>>(fa);

  AssembleArgs aa;
  memset(&aa, 0, sizeof(aa));
  aa.x = w.g.x;
  aa.dirs = dirs;
  aa.emb = emb;
  aa.grad_in = grad;
  aa.n_points = P;
  aa.n_padded = NC;
  aa.S = n_samples;
  aa.pe_degree = f->pe_code;
  aa.use_pe = f->cfg.use_position_encoding;
  aa.n_feat = f->n_feat;
  aa.nb0 = k->nb0;
  aa.nbs = k->nbs;
  aa.emb_dim = f->cfg.appearance_dim;
  aa.ref_flags = f->ref_flags;
  aa.csmall_tp = w.c.csmall;
  { ProfScope ps_(PS_ASSEMBLE, s); grad_assemble_kernel<<<(unsigned)(NC / 256 + (NC % 256 != 0)), 256, 0, s>>>(aa); }
  ColFwdArgs ca;
  memset(&ca, 0, sizeof(ca));
  fill_col_ptrs(f, packed, &ca.p, kNsCol);
  ca.feat_tp = w.c.feat;
  ca.csmall_tp = w.c.csmall;
  for (int l = 0; l < f->nlc; ++l) ca.h_tp[l] = w.c.h[l];
  ca.rgb = w.c.rgb;
  { ProfScope ps_(PS_COL_FWD, s); k->col_fwd(ca, training != 0, (unsigned)(NC / 128), s); }
  SDFHIP_CHECK_HIP(hipMemcpyAsync(rgb, w.c.rgb, (size_t)P * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (x_out != nullptr) SDFHIP_CHECK_HIP(hipMemcpyAsync(x_out, w.g.x, (size_t)P * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int sdfhip_numfield_backward(const SdfHipField* f, const float* packed, const float* level_mask, int64_t n_rays,
                                        int32_t n_samples, float delta, void* workspace, const float* sdf_bar, const float* grad_bar,
                                        const float* rgb_bar, const float* taps_bar, float* theta_bar, float* table_bar, float* emb_bar,
                                        sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(f && packed && level_mask && workspace && theta_bar && table_bar, "numfield_backward: null argument");
  SDFHIP_REQUIRE(f->ref_flags == 0, "numfield_backward: the ref-nerf colour options are built on the analytic-normal path (sdfhip_field_backward_feat)");
  if (n_rays == 0) {  // as in sdfhip_field_backward_feat: theta_bar is overwritten, and the sum over no points is 0
    SDFHIP_CHECK_HIP(hipMemsetAsync(theta_bar, 0, sizeof(float) * (size_t)f->theta_size, (hipStream_t)stream));
    return 0;
  }
  hipStream_t s = (hipStream_t)stream;
  const FieldKernels* k = f->k;
  const int64_t P = n_rays * n_samples;
  NumWs w;
  carve_num(f, P, workspace, &w);
  const int64_t NC = w.nc, N7 = w.n7;

  // 1. colour network backward: its feature cotangent lands in the centre tiles of the geometry network's featbar
  ColBwdArgs cb;
  memset(&cb, 0, sizeof(cb));
  fill_col_ptrs(f, packed, &cb.p, kNsGrad);
  cb.rgb = w.c.rgb;
  cb.rgbbar = rgb_bar;
  cb.n_points = rgb_bar != nullptr ? P : 0;
  for (int l = 0; l < f->nlc; ++l) {
    cb.h_tp[l] = w.c.h[l];
    cb.d_tp[l] = w.c.d[l];
  }
  cb.dout_tp = w.c.dout;
  cb.featbar_tp = w.c.featbar;
  cb.csmallbar_tp = w.c.csmallbar;
  { ProfScope ps_(PS_COL_BWD, s); k->col_bwd(cb, (unsigned)(NC / 128), s); }
  if (emb_bar != nullptr && f->cfg.appearance_dim > 0) {
    const int64_t ne = n_rays * f->cfg.appearance_dim;
    color_emb_reduce_kernel<<<(unsigned)((ne + 255) / 256), 256, 0, s>>>(w.c.csmallbar, k->nbs, n_rays, n_samples, f->cfg.appearance_dim,
                                                                          csmall_layout(f->ref_flags, f->cfg.appearance_dim).emb, emb_bar);
  }

  // 2. adjoint of the finite differences: sdfbar of the 7 P points
  FdArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.n_points = P;
  fa.delta = delta;
  fa.sdf_bar = sdf_bar;
  fa.grad_bar = grad_bar;
  fa.csmallbar_tp = w.c.csmallbar;
  fa.taps_bar = taps_bar;
  fa.nbs = k->nbs;
  fa.n_padded7 = N7;
  fa.sdfbar7 = w.g.sdfbar;
  fd_adjoint_kernel<<<(unsigned)((N7 + 255) / 256), 256, 0, s>>>(fa);

  // 3. geometry network backward: centre tiles with their feature cotangent, tap tiles from the sdf row alone
  GeoBwdArgs gb;
  memset(&gb, 0, sizeof(gb));
  fill_geo_ptrs(f, packed, &gb.p, kNsGrad);
  gb.featbar_tp = w.g.featbar;
  gb.sdfbar = w.g.sdfbar;
  for (int l = 0; l < f->nl; ++l) {
    gb.u_tp[l] = w.g.u[l];
    gb.zb_tp[l] = w.g.zb[l];
  }
  gb.in0bar_tp = w.g.in0bar;
  const bool split = k->geo_bwd1s != nullptr && k->has_sdf_save && N7 > NC;
  {
    ProfScope ps_(PS_GEO_BWD, s);
    if (split) {
      k->geo_bwd1(gb, (unsigned)(NC / 128), s);
      GeoBwdArgs gt = gb;
      shift_geo_bwd(f, &gt, NC / 32);
      k->geo_bwd1s(gt, (unsigned)((N7 - NC) / 128), s);
    } else {
      const int64_t fw = N7 * k->nbf * 32, cw = NC * k->nbf * 32;
      if (fw > cw) SDFHIP_CHECK_HIP(hipMemsetAsync(w.g.featbar + cw, 0, (size_t)(fw - cw) * sizeof(float), s));
      k->geo_bwd1(gb, (unsigned)(N7 / 128), s);
    }
  }

  // 4. hash-table gradient of all 7 P points (forked beside the weight-gradient GEMMs: bound by the memory-side atomic unit, not by CUs)
  GridBwdArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.grid = f->grid;
  ga.x = w.g.x;
  ga.in0bar_tp = w.g.in0bar;
  ga.mask = level_mask;
  ga.n_points = 7 * P;
  ga.pe_degree = f->pe_code;
  ga.nb0 = k->nb0;
  ga.tablebar = table_bar;
  ga.tap_points = P;
  ga.tap_delta = delta;
  const bool forked = g_side.ready();
  hipStream_t gs = s;
  if (forked) {
    SDFHIP_CHECK_HIP(hipEventRecord(g_side.fork, s));
    SDFHIP_CHECK_HIP(hipStreamWaitEvent(g_side.stream, g_side.fork, 0));
    gs = g_side.stream;
  }
  if (f->grid.n_levels > 0) {
    ProfScope ps_(PS_GRID_BWD, gs);
    const int64_t P7 = 7 * P;
    // 8 lanes per sample (7 taps + a filler) on the levels that use the tap-adjacent mapping
    if (f->grid.n_features == 8) grid_bwd8_kernel<<<dim3((unsigned)((8 * P + 255) / 256), f->grid.n_levels), 256, 0, gs>>>(ga);
    else grid_bwd_kernel<<<dim3((unsigned)((P7 + 255) / 256), f->grid.n_levels * (f->grid.n_features / 2)), 256, 0, gs>>>(ga);
  }
  if (forked) SDFHIP_CHECK_HIP(hipEventRecord(g_side.join, gs));
  notify_table_grad(f, table_bar, gs);  // the table's gradient is complete on `gs` from here on: a host may start its exchange behind it

  // 5. weight gradients.  Hidden layers: all 7 P points; the output layer's feature rows: the centre tiles only (no other point has a
  //    feature cotangent), its sdf row: all points
  WreduceBatch wb;  // one batch over both parts: w.c shares w.g's partial buffers (carve_numfield)
  begin_wreduce_batch(f, w.g, wb);
  w.c.batch = w.g.batch;
  run_geo_wgrads(f, w.g, false, N7 / 32, theta_bar, s, split ? NC / 32 : N7 / 32);
  run_col_wgrads(f, w.c, NC / 32, theta_bar, s);
  flush_wreduce(wb, s);
  if (forked) SDFHIP_CHECK_HIP(hipStreamWaitEvent(s, g_side.join, 0));
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ proposal field
static int prop_args(const SdfHipGridCfg* grid, PropArgs* a) {
  memset(a, 0, sizeof(*a));
  SDFHIP_REQUIRE(grid != nullptr && grid->n_levels == kPropLevels && grid->n_features == 2,
                 "proposal field: only 5 levels x 2 features is built");
  return make_grid_dev(grid, &a->grid);
}
static const int kPropBwdBlocks = 1024;
extern "C" int64_t sdfhip_proposal_workspace_size(void) { return (int64_t)kPropBwdBlocks * 176 * sizeof(float); }

extern "C" int sdfhip_proposal_forward(const SdfHipGridCfg* grid, const float* table, const float* w1, const float* w2,
                                       const float* origins, const float* dirs, const float* starts, const float* ends,
                                       int64_t n_rays, int32_t n_samples, int32_t contract, float* density, sdfhip_stream_t stream) {
  PropArgs a;
  const int rc = prop_args(grid, &a);
  if (rc != 0) return rc;
  SDFHIP_REQUIRE(table && w1 && w2 && origins && density && (dirs == nullptr || (starts && ends)), "proposal_forward: null argument");
  a.origins = origins;
  a.dirs = dirs;
  a.starts = starts;
  a.ends = ends;
  a.n_points = n_rays * n_samples;
  a.S = n_samples;
  a.contract = contract;
  a.table = table;
  a.w1 = w1;
  a.w2 = w2;
  a.density = density;
  if (a.n_points == 0) return 0;
  { ProfScope ps_(PS_PROP_FWD, (hipStream_t)stream); prop_fwd_kernel<<<(unsigned)((a.n_points + 255) / 256), 256, 0, (hipStream_t)stream>>>(a); }
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int sdfhip_proposal_backward(const SdfHipGridCfg* grid, const float* table, const float* w1, const float* w2,
                                        const float* origins, const float* dirs, const float* starts, const float* ends,
                                        int64_t n_rays, int32_t n_samples, int32_t contract, const float* density_bar, void* workspace,
                                        float* table_bar, float* w1_bar, float* w2_bar, sdfhip_stream_t stream) {
  PropArgs a;
  const int rc = prop_args(grid, &a);
  if (rc != 0) return rc;
  SDFHIP_REQUIRE(table && w1 && w2 && origins && (dirs == nullptr || (starts && ends)) && density_bar && workspace && table_bar && w1_bar && w2_bar,
                 "proposal_backward: null argument");
  hipStream_t s = (hipStream_t)stream;
  a.origins = origins;
  a.dirs = dirs;
  a.starts = starts;
  a.ends = ends;
  a.n_points = n_rays * n_samples;
  a.S = n_samples;
  a.contract = contract;
  a.table = table;
  a.w1 = w1;
  a.w2 = w2;
  a.densbar = density_bar;
  a.tablebar = table_bar;
  a.wpartial = (float*)workspace;
  { ProfScope ps_(PS_PROP_BWD, s); prop_bwd_kernel<<<kPropBwdBlocks, 256, 0, s>>>(a); }
  colsum_kernel<<<(176 + 31) / 32, 1024, 0, s>>>(a.wpartial, kPropBwdBlocks, 176, 176, w1_bar, 160, w2_bar);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ samplers
static int sample_spaced_impl(const float* nears, const float* fars, const float* jitter, int jitter_per_sample, int64_t n_rays,
                              int32_t n_samples, int uniform, float* bins, float* starts, float* ends, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(nears && fars && bins && starts && ends && n_samples >= 1, "sample_spaced: bad argument");
  BinsArgs a;
  a.nears = nears;
  a.fars = fars;
  a.jitter = jitter;
  a.jitter_stride = (jitter != nullptr && jitter_per_sample) ? n_samples + 1 : 0;
  a.N = (int)n_rays;
  a.S = n_samples;
  a.uniform = uniform;
  a.bins = bins;
  a.starts = starts;
  a.ends = ends;
  const int64_t total = n_rays * (n_samples + 1);
  if (total == 0) return 0;
  { ProfScope ps_(PS_SAMPLERS, (hipStream_t)stream); spaced_bins_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(a); }
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_sample_spaced(const float* nears, const float* fars, const float* jitter, int64_t n_rays, int32_t n_samples,
                                    float* bins, float* starts, float* ends, sdfhip_stream_t stream) {
  return sample_spaced_impl(nears, fars, jitter, 0, n_rays, n_samples, 0, bins, starts, ends, stream);
}
extern "C" int sdfhip_sample_uniform(const float* nears, const float* fars, const float* jitter, int32_t jitter_per_sample,
                                     int64_t n_rays, int32_t n_samples, float* bins, float* starts, float* ends,
                                     sdfhip_stream_t stream) {
  return sample_spaced_impl(nears, fars, jitter, jitter_per_sample, n_rays, n_samples, 1, bins, starts, ends, stream);
}
extern "C" int sdfhip_sample_spacing(int32_t spacing, const float* nears, const float* fars, const float* jitter,
                                     int32_t jitter_per_sample, int64_t n_rays, int32_t n_samples, float* bins, float* starts,
                                     float* ends, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(spacing >= SP_PIECEWISE && spacing <= SP_LOG, "sample_spacing: unknown spacing %d", spacing);
  return sample_spaced_impl(nears, fars, jitter, jitter_per_sample, n_rays, n_samples, spacing, bins, starts, ends, stream);
}

extern "C" int sdfhip_interlevel_terms(const float* c, const float* w, const float* cp, const float* wp, int64_t n_rays, int32_t s,
                                       int32_t s_p, float radius, float* term, float* dterm, float* w_gt, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(c && w && cp && wp && term && dterm, "interlevel_terms: null argument");
  SDFHIP_REQUIRE(s >= 1 && s_p >= 1 && 2 * (s + 1) <= 1024 && radius > 0.0f, "interlevel_terms: unsupported shape (S %d, S_p %d)", s, s_p);
  if (n_rays == 0) return 0;
  InterlevelArgs a;
  a.c = c;
  a.w = w;
  a.cp = cp;
  a.wp = wp;
  a.N = (int)n_rays;
  a.S = s;
  a.Sp = s_p;
  a.r = radius;
  a.term = term;
  a.dterm = dterm;
  a.w_gt = w_gt;
  const size_t lds = 4 * sizeof(float) * (size_t)(4 * (s + 1) + 4 * (s + 1) + (s_p + 1));
  SDFHIP_REQUIRE(lds <= 64 * 1024, "interlevel_terms: %d + %d samples do not fit the LDS staging", s, s_p);
  interlevel_kernel<<<(unsigned)((n_rays + 3) / 4), 256, lds, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

static int adam_step_impl(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int64_t step, float grad_scale, bool decoupled, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "adam_step: bad argument");
  const unsigned mis = (unsigned)(((uintptr_t)param >> 2) & 3);
  SDFHIP_REQUIRE(((uintptr_t)grad >> 2 & 3) == mis && ((uintptr_t)exp_avg >> 2 & 3) == mis && ((uintptr_t)exp_avg_sq >> 2 & 3) == mis &&
                     (uintptr_t)param % 4 == 0,
                 "adam_step: the four slices must share their offset from a 16-byte boundary (same slice of four flat buffers)");
  if (n == 0) return 0;
  AdamArgs a;
  a.head = (int32_t)std::min<int64_t>((4 - mis) & 3, n);
  a.param = param;
  a.grad = grad;
  a.exp_avg = exp_avg;
  a.exp_avg_sq = exp_avg_sq;
  a.n = n;
  // bias corrections in double on the host, as torch does with python floats (optim/adam.py _single_tensor_adam)
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  a.lr_over_bc1 = (float)((double)lr / bc1);
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  a.beta1 = beta1;
  a.beta2 = beta2;
  a.eps = eps;
  a.weight_decay = decoupled ? 0.0f : weight_decay;
  a.decay_mul = decoupled ? (float)(1.0 - (double)lr * (double)weight_decay) : 1.0f;  // python-float arithmetic, as torch's 1 - lr * weight_decay
  a.grad_scale = grad_scale;
  const int64_t n4 = (n + 3) / 4;
  const unsigned grid = (unsigned)std::min<int64_t>((n4 + 255) / 256, 256 * 16);
  adam_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int64_t step, float grad_scale, sdfhip_stream_t stream) {
  return adam_step_impl(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, false, stream);
}
extern "C" int sdfhip_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int64_t step, float grad_scale, sdfhip_stream_t stream) {
  return adam_step_impl(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, true, stream);
}

extern "C" int sdfhip_surface_root(const float* sdf, const float* starts, const float* nears, const float* fars, int64_t n_rays,
                                   int32_t n_samples, float delta, int32_t* mask, float* z, float* new_nears, float* new_fars,
                                   sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(sdf && starts && nears && fars && mask && z && new_nears && new_fars && n_samples >= 1, "surface_root: bad argument");
  if (n_rays == 0) return 0;
  RootArgs a;
  a.sdf = sdf;
  a.starts = starts;
  a.nears = nears;
  a.fars = fars;
  a.N = (int)n_rays;
  a.S = n_samples;
  a.delta = delta;
  a.mask = mask;
  a.z = z;
  a.new_nears = new_nears;
  a.new_fars = new_fars;
  surface_root_kernel<<<(unsigned)((n_rays + 255) / 256), 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- standalone hash-grid encoding (tcnn.Encoding("HashGrid") outside the fused fields)
extern "C" int sdfhip_grid_encode_forward(const SdfHipGridCfg* grid, const float* table, const float* x, int64_t n_points, float* feat,
                                          sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(grid && table && x && feat, "grid_encode_forward: null argument");
  SDFHIP_REQUIRE(grid->n_features >= 2 && grid->n_features % 2 == 0, "grid_encode: n_features must be even");
  if (n_points == 0) return 0;
  GridEncodeArgs a;
  memset(&a, 0, sizeof(a));
  const int rc = make_grid_dev(grid, &a.grid);
  if (rc != 0) return rc;
  a.x = x;
  a.n_points = n_points;
  a.table = table;
  a.feat = feat;
  grid_encode_kernel<false><<<dim3((unsigned)((n_points + 255) / 256), grid->n_levels * (grid->n_features / 2)), 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_grid_encode_backward(const SdfHipGridCfg* grid, const float* x, int64_t n_points, const float* feat_bar,
                                           float* table_bar, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(grid && x && feat_bar && table_bar, "grid_encode_backward: null argument");
  SDFHIP_REQUIRE(grid->n_features >= 2 && grid->n_features % 2 == 0, "grid_encode: n_features must be even");
  if (n_points == 0) return 0;
  GridEncodeArgs a;
  memset(&a, 0, sizeof(a));
  const int rc = make_grid_dev(grid, &a.grid);
  if (rc != 0) return rc;
  a.x = x;
  a.n_points = n_points;
  a.featbar = feat_bar;
  a.tablebar = table_bar;
  grid_encode_kernel<true><<<dim3((unsigned)((n_points + 255) / 256), grid->n_levels * (grid->n_features / 2)), 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int sdfhip_grid_cell_dump(const SdfHipGridCfg* grid, const float* x, int64_t n_points, uint32_t* idx, float* w,
                                    sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(grid && x && idx && w, "grid_cell_dump: null argument");
  if (n_points == 0) return 0;
  GridDumpArgs a;
  memset(&a, 0, sizeof(a));
  const int rc = make_grid_dev(grid, &a.grid);
  if (rc != 0) return rc;
  a.x = x;
  a.n_points = n_points;
  a.idx = idx;
  a.w = w;
  grid_cell_dump_kernel<<<dim3((unsigned)((n_points + 255) / 256), grid->n_levels), 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- weight-normalised parameters -> theta (and back), scalar losses of the surface models (theta_kernels.h)
static int fill_theta_layers(const SdfHipField* f, const float* const* v, const float* const* g, const float* const* b, int32_t n_lin,
                             ThetaLayers* L) {
  SDFHIP_REQUIRE(f && v && g && b, "weightnorm_theta: null argument");
  SDFHIP_REQUIRE(n_lin == (int32_t)f->lin.size() && n_lin <= kThetaMaxLin, "weightnorm_theta: %d layers given, the field has %d (max %d)", n_lin,
                 (int)f->lin.size(), kThetaMaxLin);
  memset(L, 0, sizeof(*L));
  L->n_lin = n_lin;
  int rows = 0;
  for (int l = 0; l < n_lin; ++l) {
    SDFHIP_REQUIRE(v[l] && g[l] && b[l], "weightnorm_theta: null parameter pointer for layer %d", l);
    L->row_start[l] = rows;
    L->out_dim[l] = f->lin[l].out_dim;
    L->in_dim[l] = f->lin[l].in_dim;
    L->w_off[l] = f->lin[l].w_off;
    L->b_off[l] = f->lin[l].b_off;
    L->v[l] = v[l];
    L->g[l] = g[l];
    L->b[l] = b[l];
    rows += f->lin[l].out_dim;
  }
  L->row_start[n_lin] = rows;
  L->total_rows = rows;
  return 0;
}
extern "C" int64_t sdfhip_field_weightnorm_rows(const SdfHipField* f) {
  int64_t rows = 0;
  for (const LinearInfo& li : f->lin) rows += li.out_dim;
  return rows;
}
extern "C" int sdfhip_field_theta_from_weightnorm(const SdfHipField* f, const float* const* v, const float* const* g, const float* const* b,
                                                  int32_t n_lin, float* theta, float* inv_norm, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(theta && inv_norm, "theta_from_weightnorm: null output");
  ThetaLayers L;
  const int rc = fill_theta_layers(f, v, g, b, n_lin, &L);
  if (rc != 0) return rc;
  weightnorm_theta_fwd_kernel<<<(unsigned)((L.total_rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(L, theta, inv_norm);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_field_theta_backward_weightnorm(const SdfHipField* f, const float* const* v, const float* const* g, const float* const* b,
                                                      int32_t n_lin, const float* inv_norm, const float* theta_bar, float* const* v_bar,
                                                      float* const* g_bar, float* const* b_bar, int32_t accumulate, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(theta_bar && inv_norm && v_bar && g_bar && b_bar, "theta_backward_weightnorm: null argument");
  ThetaLayers L;
  const int rc = fill_theta_layers(f, v, g, b, n_lin, &L);
  if (rc != 0) return rc;
  for (int l = 0; l < n_lin; ++l) {
    L.v_bar[l] = v_bar[l];
    L.g_bar[l] = g_bar[l];
    L.b_bar[l] = b_bar[l];
  }
  weightnorm_theta_bwd_kernel<<<(unsigned)((L.total_rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(L, theta_bar, inv_norm, accumulate);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// MonoSDF depth prior (ScaleAndShiftInvariantLoss as base_surface_model.py:427-437 calls it) and the foreground-mask BCE (:415-420)
static int fill_depth_loss(DepthLossArgs* a, const float* pred, const float* gt, int64_t n, int32_t rows, float gt_scale, float gt_shift,
                           float alpha) {
  SDFHIP_REQUIRE(pred && gt && n >= 1 && rows >= 1 && n % rows == 0 && n < (1 << 30), "mono_depth_loss: %lld rays do not form a %d-row image",
                 (long long)n, rows);
  memset(a, 0, sizeof(*a));
  a->pred = pred;
  a->gt = gt;
  a->n = (int)n;
  a->rows = rows;
  a->width = (int)(n / rows);
  a->gt_scale = gt_scale;
  a->gt_shift = gt_shift;
  a->alpha = alpha;
  return 0;
}
extern "C" int sdfhip_mono_depth_loss_forward(const float* depth_pred, const float* depth_gt, int64_t n_rays, int32_t rows, float gt_scale,
                                              float gt_shift, float alpha, float* loss, float* state10, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(loss && state10, "mono_depth_loss_forward: null output");
  DepthLossArgs a;
  if (int rc = fill_depth_loss(&a, depth_pred, depth_gt, n_rays, rows, gt_scale, gt_shift, alpha)) return rc;
  a.loss = loss;
  a.state = state10;
  depth_loss_fwd_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_mono_depth_loss_backward(const float* depth_pred, const float* depth_gt, int64_t n_rays, int32_t rows, float gt_scale,
                                               float gt_shift, float alpha, const float* state10, const float* loss_bar, float* pred_bar,
                                               sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(state10 && loss_bar && pred_bar, "mono_depth_loss_backward: null argument");
  DepthLossArgs a;
  if (int rc = fill_depth_loss(&a, depth_pred, depth_gt, n_rays, rows, gt_scale, gt_shift, alpha)) return rc;
  a.state = const_cast<float*>(state10);
  a.loss_bar = loss_bar;
  a.pred_bar = pred_bar;
  depth_loss_bwd_kernel<<<(unsigned)std::min<int64_t>((n_rays + 255) / 256, 1024), 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_fg_mask_loss_forward(const float* acc, const float* label, int64_t n_rays, float mult, float* loss, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(acc && label && loss && n_rays >= 1 && n_rays < (1 << 30), "fg_mask_loss_forward: bad argument");
  FgLossArgs a;
  memset(&a, 0, sizeof(a));
  a.acc = acc;
  a.label = label;
  a.n = (int)n_rays;
  a.scale = mult / (float)n_rays;
  a.loss = loss;
  fg_loss_fwd_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_fg_mask_loss_backward(const float* acc, const float* label, int64_t n_rays, float mult, const float* loss_bar,
                                            float* acc_bar, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(acc && label && loss_bar && acc_bar && n_rays >= 1 && n_rays < (1 << 30), "fg_mask_loss_backward: bad argument");
  FgLossArgs a;
  memset(&a, 0, sizeof(a));
  a.acc = acc;
  a.label = label;
  a.n = (int)n_rays;
  a.scale = mult / (float)n_rays;
  a.loss_bar = loss_bar;
  a.acc_bar = acc_bar;
  fg_loss_bwd_kernel<<<(unsigned)std::min<int64_t>((n_rays + 255) / 256, 1024), 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

static const int kSensorDepthBlocks = 512;
static int fill_sensor_depth(SensorDepthArgs* a, const float* depth_pred, const float* depth_gt, const float* sdf, const float* starts,
                             const float* directions_norm, int64_t n_rays, int64_t n_samples, float truncation) {
  SDFHIP_REQUIRE(depth_pred && depth_gt && sdf && starts, "sensor_depth_loss: null argument");
  SDFHIP_REQUIRE(n_rays >= 1 && n_samples >= 1 && n_rays < (1 << 30) && n_samples < (1 << 20), "sensor_depth_loss: bad shape");
  memset(a, 0, sizeof(*a));
  a->depth_pred = depth_pred;
  a->depth_gt = depth_gt;
  a->sdf = sdf;
  a->starts = starts;
  a->dnorm = directions_norm;
  a->n_rays = (int32_t)n_rays;
  a->n_samples = (int32_t)n_samples;
  a->truncation = truncation;
  a->n_blocks = (int32_t)std::min<int64_t>((n_rays * n_samples + 255) / 256, kSensorDepthBlocks);
  return 0;
}
extern "C" size_t sdfhip_sensor_depth_loss_workspace_size(void) { return (size_t)kSensorDepthBlocks * 6 * sizeof(double); }
extern "C" int sdfhip_sensor_depth_loss_forward(const float* depth_pred, const float* depth_gt, const float* sdf, const float* starts,
                                                const float* directions_norm, int64_t n_rays, int64_t n_samples, float truncation,
                                                void* workspace, float* losses3, float* state4, sdfhip_stream_t stream) {
  SensorDepthArgs a;
  if (int rc = fill_sensor_depth(&a, depth_pred, depth_gt, sdf, starts, directions_norm, n_rays, n_samples, truncation)) return rc;
  SDFHIP_REQUIRE(workspace && losses3 && state4, "sensor_depth_loss_forward: null argument");
  a.partial = (double*)workspace;
  a.losses = losses3;
  a.state = state4;
  sensor_depth_partial_kernel<<<(unsigned)a.n_blocks, 256, 0, (hipStream_t)stream>>>(a);
  sensor_depth_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_sensor_depth_loss_backward(const float* depth_pred, const float* depth_gt, const float* sdf, const float* starts,
                                                 const float* directions_norm, int64_t n_rays, int64_t n_samples, float truncation,
                                                 const float* state4, const float* losses_bar3, float* sdf_bar, float* depth_bar,
                                                 sdfhip_stream_t stream) {
  SensorDepthArgs a;
  if (int rc = fill_sensor_depth(&a, depth_pred, depth_gt, sdf, starts, directions_norm, n_rays, n_samples, truncation)) return rc;
  SDFHIP_REQUIRE(state4 && losses_bar3 && sdf_bar && depth_bar, "sensor_depth_loss_backward: null argument");
  a.state = const_cast<float*>(state4);
  a.losses_bar = losses_bar3;
  a.sdf_bar = sdf_bar;
  a.depth_bar = depth_bar;
  sensor_depth_bwd_kernel<<<(unsigned)std::min<int64_t>((n_rays * n_samples + 255) / 256, 4096), 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

static const int kSurfaceLossBlocks = 512;
extern "C" int64_t sdfhip_surface_loss_workspace_floats(void) { return (int64_t)kSurfaceLossBlocks * SL_COUNT; }
static void fill_surface_loss(SurfaceLossArgs* a, const float* rgb, const float* image, int64_t n_rays, const float* grad, const float* sdf,
                              const float* taps, float delta, int64_t n_points, const float* n_pred, const float* n_gt, const float* scale4) {
  memset(a, 0, sizeof(*a));
  a->rgb = rgb;
  a->image = image;
  a->grad = grad;
  a->sdf = sdf;
  a->taps = taps;
  a->n_pred = n_pred;
  a->n_gt = n_gt;
  a->n_rays = n_rays;
  a->n_points = (grad || taps) ? n_points : 0;
  a->inv_delta2 = taps ? 1.0f / (delta * delta) : 0.0f;
  for (int k = 0; k < SL_COUNT; ++k) a->scale[k] = scale4[k];
}
extern "C" int sdfhip_surface_loss_forward(const float* rgb, const float* image, int64_t n_rays, const float* grad, const float* sdf,
                                           const float* taps, float delta, int64_t n_points, const float* n_pred, const float* n_gt,
                                           const float* scale4_host, float* workspace, float* loss4, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(rgb && image && scale4_host && workspace && loss4 && n_rays >= 1, "surface_loss_forward: bad argument");
  SDFHIP_REQUIRE((taps == nullptr) == (sdf == nullptr) && (n_pred == nullptr) == (n_gt == nullptr), "surface_loss_forward: taps need sdf, n_pred needs n_gt");
  SurfaceLossArgs a;
  fill_surface_loss(&a, rgb, image, n_rays, grad, sdf, taps, delta, n_points, n_pred, n_gt, scale4_host);
  const int64_t items = std::max<int64_t>(n_rays, a.n_points);
  a.n_blocks = (int)std::min<int64_t>(kSurfaceLossBlocks, (items + 255) / 256);
  a.partial = workspace;
  a.loss = loss4;
  surface_loss_partial_kernel<<<(unsigned)a.n_blocks, 256, 0, (hipStream_t)stream>>>(a);
  surface_loss_finish_kernel<<<1, 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_surface_loss_backward(const float* rgb, const float* image, int64_t n_rays, const float* grad, const float* sdf,
                                            const float* taps, float delta, int64_t n_points, const float* n_pred, const float* n_gt,
                                            const float* scale4_host, const float* const* loss_bar4, float* rgb_bar, float* grad_bar,
                                            float* sdf_bar, float* taps_bar, float* n_pred_bar, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(rgb && image && scale4_host && loss_bar4 && n_rays >= 1, "surface_loss_backward: bad argument");
  SDFHIP_REQUIRE((grad_bar == nullptr || grad) && (taps_bar == nullptr || (taps && sdf && sdf_bar)) && (n_pred_bar == nullptr || (n_pred && n_gt)),
                 "surface_loss_backward: an output gradient was asked for without its input");
  SurfaceLossArgs a;
  fill_surface_loss(&a, rgb, image, n_rays, grad, sdf, taps, delta, n_points, n_pred, n_gt, scale4_host);
  for (int k = 0; k < SL_COUNT; ++k) a.loss_bar[k] = loss_bar4[k];
  a.rgb_bar = rgb_bar;
  a.grad_bar = grad_bar;
  a.sdf_bar = sdf_bar;
  a.taps_bar = taps_bar;
  a.n_pred_bar = n_pred_bar;
  const int64_t items = std::max<int64_t>(n_rays, a.n_points);
  surface_loss_bwd_kernel<<<(unsigned)std::min<int64_t>(4096, (items + 255) / 256), 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- packed-sample path (NeuS-acc): occupancy-grid marching, segmented compositing
static void fill_march(MarchArgs* a, const float* origins, const float* dirs, const float* t_min, const float* t_max, const float* roi_aabb6,
                       const uint8_t* binary, int64_t n_rays, int32_t resolution, float step) {
  memset(a, 0, sizeof(*a));
  a->origins = origins;
  a->dirs = dirs;
  a->t_min = t_min;
  a->t_max = t_max;
  a->binary = binary;
  for (int k = 0; k < 3; ++k) {
    a->roi_min[k] = roi_aabb6[k];
    a->roi_max[k] = roi_aabb6[3 + k];
  }
  a->N = (int)n_rays;
  a->R = resolution;
  a->step = step;
  a->step_dev = nullptr;
  a->capacity = -1;
}
extern "C" int sdfhip_march_count(const float* origins, const float* dirs, const float* t_min, const float* t_max, const float* roi_aabb6_host,
                                  const uint8_t* binary, int64_t n_rays, int32_t resolution, float step, int32_t* counts,
                                  sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(origins && dirs && t_min && t_max && roi_aabb6_host && binary && counts && resolution >= 1 && step > 0.0f,
                 "march_count: bad argument");
  if (n_rays == 0) return 0;
  MarchArgs a;
  fill_march(&a, origins, dirs, t_min, t_max, roi_aabb6_host, binary, n_rays, resolution, step);
  a.counts = counts;
  march_kernel<false><<<(unsigned)((n_rays + 127) / 128), 128, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_march_write(const float* origins, const float* dirs, const float* t_min, const float* t_max, const float* roi_aabb6_host,
                                  const uint8_t* binary, int64_t n_rays, int32_t resolution, float step, const int64_t* offsets,
                                  int64_t* ray_indices, float* t_starts, float* t_ends, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(origins && dirs && t_min && t_max && roi_aabb6_host && binary && offsets && ray_indices && t_starts && t_ends &&
                     resolution >= 1 && step > 0.0f, "march_write: bad argument");
  if (n_rays == 0) return 0;
  MarchArgs a;
  fill_march(&a, origins, dirs, t_min, t_max, roi_aabb6_host, binary, n_rays, resolution, step);
  a.offsets = offsets;
  a.ray_indices = ray_indices;
  a.t_starts = t_starts;
  a.t_ends = t_ends;
  march_kernel<true><<<(unsigned)((n_rays + 127) / 128), 128, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
// The same two passes for a host that never reads the sample count back: the step may be a DEVICE scalar (step_dev,
// overrides `step`), and the write pass drops what falls behind `capacity` packed samples (< 0: unbounded) - the caller sizes its arrays by
// a bound, clamps (offsets, counts) to it on the device and checks an overflow flag now and then (ray_samplers.py: march_occupancy_grid).
extern "C" int sdfhip_march_count_dev(const float* origins, const float* dirs, const float* t_min, const float* t_max,
                                      const float* roi_aabb6_host, const uint8_t* binary, int64_t n_rays, int32_t resolution, float step,
                                      const float* step_dev, int32_t* counts, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(origins && dirs && t_min && t_max && roi_aabb6_host && binary && counts && resolution >= 1 && (step > 0.0f || step_dev),
                 "march_count_dev: bad argument");
  if (n_rays == 0) return 0;
  MarchArgs a;
  fill_march(&a, origins, dirs, t_min, t_max, roi_aabb6_host, binary, n_rays, resolution, step);
  a.step_dev = step_dev;
  a.counts = counts;
  march_kernel<false><<<(unsigned)((n_rays + 127) / 128), 128, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_march_write_capped(const float* origins, const float* dirs, const float* t_min, const float* t_max,
                                         const float* roi_aabb6_host, const uint8_t* binary, int64_t n_rays, int32_t resolution, float step,
                                         const float* step_dev, const int64_t* offsets, int64_t capacity, int64_t* ray_indices,
                                         float* t_starts, float* t_ends, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(origins && dirs && t_min && t_max && roi_aabb6_host && binary && offsets && ray_indices && t_starts && t_ends &&
                     resolution >= 1 && (step > 0.0f || step_dev), "march_write_capped: bad argument");
  if (n_rays == 0) return 0;
  MarchArgs a;
  fill_march(&a, origins, dirs, t_min, t_max, roi_aabb6_host, binary, n_rays, resolution, step);
  a.step_dev = step_dev;
  a.capacity = capacity;
  a.offsets = offsets;
  a.ray_indices = ray_indices;
  a.t_starts = t_starts;
  a.t_ends = t_ends;
  march_kernel<true><<<(unsigned)((n_rays + 127) / 128), 128, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_packed_resample(const float* t_starts, const float* t_ends, const float* weights, const int64_t* offsets,
                                      const int32_t* counts, int64_t n_rays, int32_t n_out, const int64_t* out_offsets, float* out_starts,
                                      float* out_ends, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(t_starts && t_ends && weights && offsets && counts && out_offsets && out_starts && out_ends && n_out >= 1,
                 "packed_resample: bad argument");
  if (n_rays == 0) return 0;
  ResampleArgs a;
  memset(&a, 0, sizeof(a));
  a.offsets = offsets;
  a.counts = counts;
  a.t_starts = t_starts;
  a.t_ends = t_ends;
  a.weights = weights;
  a.N = (int)n_rays;
  a.n_out = n_out;
  a.out_starts = out_starts;
  a.out_ends = out_ends;
  a.out_offsets = out_offsets;
  packed_resample_kernel<<<(unsigned)((n_rays + 127) / 128), 128, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_packed_weights_forward(const float* alpha, const int64_t* offsets, const int32_t* counts, int64_t n_rays,
                                             float* weights, float* trans, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(offsets && counts && weights && trans, "packed_weights_forward: null argument");
  if (n_rays == 0) return 0;
  PackedArgs a;
  memset(&a, 0, sizeof(a));
  a.offsets = offsets;
  a.counts = counts;
  a.N = (int)n_rays;
  a.alpha = alpha;
  a.weights = weights;
  a.trans = trans;
  packed_weights_fwd_kernel<<<(unsigned)((n_rays + 3) / 4), 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_packed_weights_backward(const float* alpha, const float* weights, const float* trans, const float* weights_bar,
                                              const int64_t* offsets, const int32_t* counts, int64_t n_rays, float* alpha_bar,
                                              sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(offsets && counts && alpha_bar, "packed_weights_backward: null argument");
  if (n_rays == 0) return 0;
  PackedArgs a;
  memset(&a, 0, sizeof(a));
  a.offsets = offsets;
  a.counts = counts;
  a.N = (int)n_rays;
  a.alpha = alpha;
  a.weights = const_cast<float*>(weights);
  a.trans = const_cast<float*>(trans);
  a.wbar = weights_bar;
  a.alphabar = alpha_bar;
  packed_weights_bwd_kernel<<<(unsigned)((n_rays + 3) / 4), 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_packed_accumulate(const float* weights, const float* values, const int64_t* offsets, const int32_t* counts,
                                        int64_t n_rays, int32_t dim, float* out, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(offsets && counts && out && dim >= 1, "packed_accumulate: bad argument");
  if (n_rays == 0) return 0;
  PackedArgs a;
  memset(&a, 0, sizeof(a));
  a.offsets = offsets;
  a.counts = counts;
  a.N = (int)n_rays;
  a.D = dim;
  a.weights = const_cast<float*>(weights);
  a.values = values;
  a.out = out;
  packed_accumulate_kernel<<<(unsigned)((n_rays + 3) / 4), 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

#define SDFHIP_DISPATCH_C(S, CALL)                                      \
  do {                                                                  \
    const int c_ = ((S) + 63) / 64;                                     \
    if (c_ <= 1) { constexpr int C = 1; CALL; }                         \
    else if (c_ <= 2) { constexpr int C = 2; CALL; }                    \
    else if (c_ <= 4) { constexpr int C = 4; CALL; }                    \
    else if (c_ <= 8) { constexpr int C = 8; CALL; }                    \
    else { constexpr int C = 16; CALL; }                                \
  } while (0)

static int sample_pdf_impl(const float* weights, const float* bins_in, const float* nears, const float* fars, const float* jitter,
                           int jitter_per_sample, int uniform, int64_t n_rays, int32_t s_in, int32_t s_out, float anneal,
                           float histogram_padding, float* bins_out, float* starts, float* ends, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(weights && bins_in && nears && fars && bins_out && starts && ends, "sample_pdf: null argument");
  SDFHIP_REQUIRE(s_in >= 1 && s_in <= 64 * kMaxPerLane && s_out >= 1, "sample_pdf: unsupported sample counts (%d -> %d)", s_in, s_out);
  PdfArgs a;
  a.weights = weights;
  a.bins_in = bins_in;
  a.nears = nears;
  a.fars = fars;
  a.jitter = jitter;
  a.jitter_stride = (jitter != nullptr && jitter_per_sample) ? s_out + 1 : 0;
  a.uniform = uniform;
  a.N = (int)n_rays;
  a.S_in = s_in;
  a.S_out = s_out;
  a.anneal = anneal;
  a.histogram_padding = histogram_padding;
  a.eps = 1e-5f;
  const int nbins = s_out + 1;
  a.u_end = (float)(1.0 - 1.0 / (double)nbins);     // python float arithmetic, then cast (torch.linspace end)
  a.u_center = (float)(1.0 / (2.0 * (double)nbins));
  a.bins_out = bins_out;
  a.starts = starts;
  a.ends = ends;
  if (n_rays == 0) return 0;
  const unsigned grid = (unsigned)((n_rays + 3) / 4);
  { ProfScope ps_(PS_SAMPLERS, (hipStream_t)stream); SDFHIP_DISPATCH_C(s_in, (pdf_sample_kernel<C><<<grid, 256, 0, (hipStream_t)stream>>>(a))); }
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_sample_pdf(const float* weights, const float* bins_in, const float* nears, const float* fars, const float* jitter,
                                 int64_t n_rays, int32_t s_in, int32_t s_out, float anneal, float histogram_padding, float* bins_out,
                                 float* starts, float* ends, sdfhip_stream_t stream) {
  return sample_pdf_impl(weights, bins_in, nears, fars, jitter, 0, 0, n_rays, s_in, s_out, anneal, histogram_padding, bins_out, starts, ends,
                         stream);
}
extern "C" int sdfhip_sample_pdf_uniform(const float* weights, const float* bins_in, const float* nears, const float* fars,
                                         const float* jitter, int32_t jitter_per_sample, int64_t n_rays, int32_t s_in, int32_t s_out,
                                         float histogram_padding, float* bins_out, float* starts, float* ends, sdfhip_stream_t stream) {
  return sample_pdf_impl(weights, bins_in, nears, fars, jitter, jitter_per_sample, 1, n_rays, s_in, s_out, 1.0f, histogram_padding, bins_out,
                         starts, ends, stream);
}

extern "C" int sdfhip_sample_pdf_spacing(int32_t spacing, const float* weights, const float* bins_in, const float* nears, const float* fars,
                                         const float* jitter, int32_t jitter_per_sample, int64_t n_rays, int32_t s_in, int32_t s_out,
                                         float anneal, float histogram_padding, float* bins_out, float* starts, float* ends,
                                         sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(spacing >= SP_PIECEWISE && spacing <= SP_LOG, "sample_pdf_spacing: unknown spacing %d", spacing);
  return sample_pdf_impl(weights, bins_in, nears, fars, jitter, jitter_per_sample, spacing, n_rays, s_in, s_out, anneal, histogram_padding,
                         bins_out, starts, ends, stream);
}

extern "C" int sdfhip_merge_uniform(const float* bins_1, const float* bins_2, const float* nears, const float* fars, int64_t n_rays,
                                    int32_t s1, int32_t s2, float* merged_bins, int32_t* merged_index, float* merged_starts,
                                    float* merged_ends, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(bins_1 && bins_2 && nears && fars && merged_bins && merged_index && merged_starts && merged_ends && s1 >= 1 && s2 >= 1,
                 "merge_uniform: bad argument");
  if (n_rays == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  MergeArgs a;
  a.bins_1 = bins_1;
  a.bins_2 = bins_2;
  a.nears = nears;
  a.fars = fars;
  a.N = (int)n_rays;
  a.S1 = s1;
  a.S2 = s2;
  a.merged_bins = merged_bins;
  a.merged_index = merged_index;
  const int M = s1 + s2;
  const int64_t total = n_rays * (M + 1);
  ProfScope ps_(PS_SAMPLERS, s);
  merge_bins_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(a);
  uniform_euclid_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(merged_bins, nears, fars, (int)n_rays, M, merged_starts, merged_ends);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int sdfhip_volsdf_bound_step(const float* bins_in, const float* sdf_a, const float* sdf_b, const int32_t* index,
                                        const float* nears, const float* fars, const float* beta_in, const float* beta0, int64_t n_rays,
                                        int32_t s_a, int32_t s_b, float eps, int32_t beta_iters, float* sdf_merged, float* beta_out,
                                        float* weights, float* err_weights, int32_t* not_converged, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(bins_in && sdf_a && nears && fars && beta_in && beta0 && sdf_merged && beta_out && weights && err_weights && not_converged,
                 "volsdf_bound_step: null argument");
  SDFHIP_REQUIRE((s_b == 0) == (index == nullptr) && (s_b == 0 || sdf_b != nullptr), "volsdf_bound_step: sdf_b / index must come together");
  const int S = s_a + s_b;
  SDFHIP_REQUIRE(S >= 2 && S <= 64 * kMaxPerLane, "volsdf_bound_step: %d samples unsupported (2..%d)", S, 64 * kMaxPerLane);
  VolsdfStepArgs a;
  a.bins_in = bins_in;
  a.sdf_a = sdf_a;
  a.sdf_b = sdf_b;
  a.index = index;
  a.nears = nears;
  a.fars = fars;
  a.beta_in = beta_in;
  a.beta0 = beta0;
  a.N = (int)n_rays;
  a.Sa = s_a;
  a.Sb = s_b;
  a.beta_iters = beta_iters;
  a.eps = eps;
  a.sdf_merged = sdf_merged;
  a.beta_out = beta_out;
  a.weights = weights;
  a.err_weights = err_weights;
  a.not_converged = not_converged;
  if (n_rays == 0) return 0;
  const unsigned grid = (unsigned)((n_rays + 3) / 4);
  { ProfScope ps_(PS_SAMPLERS, (hipStream_t)stream); SDFHIP_DISPATCH_C(S, (volsdf_step_kernel<C><<<grid, 256, 0, (hipStream_t)stream>>>(a))); }
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int sdfhip_neus_upsample(const float* bins_in, const float* sdf_a, const float* sdf_b, const int32_t* index, const float* nears,
                                    const float* fars, const float* jitter, int32_t jitter_per_sample, int64_t n_rays, int32_t s_a,
                                    int32_t s_b, int32_t n_new, float inv_s, float* sdf_merged, float* new_bins, float* new_starts, float* new_ends,
                                    float* merged_bins, int32_t* merged_index, float* merged_starts, float* merged_ends,
                                    sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(bins_in && sdf_a && nears && fars && sdf_merged && new_bins && new_starts && new_ends && merged_bins && merged_index &&
                     merged_starts && merged_ends,
                 "neus_upsample: null argument");
  SDFHIP_REQUIRE((s_b == 0) == (index == nullptr) && (s_b == 0 || sdf_b != nullptr), "neus_upsample: sdf_b / index must come together");
  const int S = s_a + s_b;
  SDFHIP_REQUIRE(S >= 2 && n_new >= 1 && n_new <= kNeusUpMaxNew && S + n_new + 1 <= 64 * kMaxPerLane,
                 "neus_upsample: %d + %d samples unsupported (new <= %d, total <= %d)", S, n_new, kNeusUpMaxNew, 64 * kMaxPerLane - 1);
  NeusUpArgs a;
  a.bins_in = bins_in;
  a.sdf_a = sdf_a;
  a.sdf_b = sdf_b;
  a.index = index;
  a.nears = nears;
  a.fars = fars;
  a.jitter = jitter;
  a.jitter_stride = jitter != nullptr && jitter_per_sample ? n_new + 1 : 0;
  a.N = (int)n_rays;
  a.Sa = s_a;
  a.Sb = s_b;
  a.n_new = n_new;
  a.inv_s = inv_s;
  a.histogram_padding = 1e-5f;  // NeuSSampler's PDFSampler (ray_samplers.py:843-847)
  a.eps = 1e-5f;
  const int nbins = n_new + 1;
  a.u_end = (float)(1.0 - 1.0 / (double)nbins);
  a.u_center = (float)(1.0 / (2.0 * (double)nbins));
  a.sdf_merged = sdf_merged;
  a.new_bins = new_bins;
  a.new_starts = new_starts;
  a.new_ends = new_ends;
  a.merged_bins = merged_bins;
  a.merged_index = merged_index;
  a.merged_starts = merged_starts;
  a.merged_ends = merged_ends;
  if (n_rays == 0) return 0;
  const unsigned grid = (unsigned)((n_rays + 3) / 4);
  { ProfScope ps_(PS_SAMPLERS, (hipStream_t)stream); SDFHIP_DISPATCH_C(S + n_new + 1, (neus_upsample_kernel<C><<<grid, 256, 0, (hipStream_t)stream>>>(a))); }
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ weights + renderers
extern "C" int sdfhip_density_weights_forward(const float* density, const float* starts, const float* ends, int64_t n_rays,
                                              int32_t n_samples, float* weights, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(density && starts && ends && weights, "density_weights_forward: null argument");
  SDFHIP_REQUIRE(n_samples >= 1 && n_samples <= 64 * kMaxPerLane, "density_weights: n_samples %d unsupported", n_samples);
  DensityWeightsArgs a;
  memset(&a, 0, sizeof(a));
  a.density = density;
  a.starts = starts;
  a.ends = ends;
  a.N = (int)n_rays;
  a.S = n_samples;
  a.weights = weights;
  if (n_rays == 0) return 0;
  const unsigned grid = (unsigned)((n_rays + 3) / 4);
  { ProfScope ps_(PS_DENSITY_W, (hipStream_t)stream); SDFHIP_DISPATCH_C(n_samples, (density_weights_fwd_kernel<C><<<grid, 256, 0, (hipStream_t)stream>>>(a))); }
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int sdfhip_density_weights_backward(const float* density, const float* starts, const float* ends, int64_t n_rays,
                                               int32_t n_samples, const float* weights_bar, float* density_bar,
                                               sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(density && starts && ends && weights_bar && density_bar, "density_weights_backward: null argument");
  SDFHIP_REQUIRE(n_samples >= 1 && n_samples <= 64 * kMaxPerLane, "density_weights: n_samples %d unsupported", n_samples);
  DensityWeightsArgs a;
  memset(&a, 0, sizeof(a));
  a.density = density;
  a.starts = starts;
  a.ends = ends;
  a.N = (int)n_rays;
  a.S = n_samples;
  a.weightsbar = weights_bar;
  a.densitybar = density_bar;
  if (n_rays == 0) return 0;
  const unsigned grid = (unsigned)((n_rays + 3) / 4);
  { ProfScope ps_(PS_DENSITY_W, (hipStream_t)stream); SDFHIP_DISPATCH_C(n_samples, (density_weights_bwd_kernel<C><<<grid, 256, 0, (hipStream_t)stream>>>(a))); }
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ void minmax_init_kernel(float* mm) {
  mm[0] = __uint_as_float(0x7f800000u);
  mm[1] = 0.0f;
}

static int neus_render_forward_impl(const float* sdf, const float* grad, const float* rgb, const float* dirs, const float* starts,
                                    const float* ends, const float* variance, const float* background, float cos_anneal,
                                    int64_t n_rays, int32_t n_samples, float* alpha, float* weights, float* out_rgb,
                                    float* out_depth_raw, float* out_depth, float* out_normal, float* out_acc,
                                    float* steps_minmax, const float* origins, const float* bg_density, const float* bg_rgb,
                                    float* rgb_merged, sdfhip_stream_t stream);
extern "C" int sdfhip_neus_render_forward(const float* sdf, const float* grad, const float* rgb, const float* dirs, const float* starts,
                                          const float* ends, const float* variance, const float* background, float cos_anneal,
                                          int64_t n_rays, int32_t n_samples, float* alpha, float* weights, float* out_rgb,
                                          float* out_depth_raw, float* out_depth, float* out_normal, float* out_acc,
                                          float* steps_minmax, sdfhip_stream_t stream) {
  return neus_render_forward_impl(sdf, grad, rgb, dirs, starts, ends, variance, background, cos_anneal, n_rays, n_samples, alpha, weights,
                                  out_rgb, out_depth_raw, out_depth, out_normal, out_acc, steps_minmax, nullptr, nullptr, nullptr, nullptr, stream);
}
extern "C" int sdfhip_neus_render_bg_forward(const float* sdf, const float* grad, const float* rgb, const float* dirs, const float* starts,
                                             const float* ends, const float* variance, const float* background, float cos_anneal,
                                             int64_t n_rays, int32_t n_samples, const float* origins, const float* bg_density,
                                             const float* bg_rgb, float* alpha, float* weights, float* out_rgb, float* out_depth_raw,
                                             float* out_depth, float* out_normal, float* out_acc, float* steps_minmax,
                                             float* rgb_merged, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(origins && bg_density && bg_rgb, "neus_render_bg_forward: null argument");
  return neus_render_forward_impl(sdf, grad, rgb, dirs, starts, ends, variance, background, cos_anneal, n_rays, n_samples, alpha, weights,
                                  out_rgb, out_depth_raw, out_depth, out_normal, out_acc, steps_minmax, origins, bg_density, bg_rgb, rgb_merged,
                                  stream);
}
static int neus_render_forward_impl(const float* sdf, const float* grad, const float* rgb, const float* dirs, const float* starts,
                                    const float* ends, const float* variance, const float* background, float cos_anneal,
                                    int64_t n_rays, int32_t n_samples, float* alpha, float* weights, float* out_rgb,
                                    float* out_depth_raw, float* out_depth, float* out_normal, float* out_acc,
                                    float* steps_minmax, const float* origins, const float* bg_density, const float* bg_rgb,
                                    float* rgb_merged, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(sdf && grad && rgb && dirs && starts && ends && variance && alpha && weights && out_rgb && out_depth_raw && out_depth &&
                     out_normal && out_acc && steps_minmax,
                 "neus_render_forward: null argument");
  SDFHIP_REQUIRE(n_samples >= 1 && n_samples <= 64 * kMaxPerLane, "neus_render: n_samples %d unsupported", n_samples);
  hipStream_t s = (hipStream_t)stream;
  NeusRenderArgs a;
  memset(&a, 0, sizeof(a));
  a.sdf = sdf;
  a.grad = grad;
  a.rgb = rgb;
  a.dirs = dirs;
  a.starts = starts;
  a.ends = ends;
  a.variance = variance;
  a.bg = background;
  a.cos_anneal = cos_anneal;
  a.N = (int)n_rays;
  a.S = n_samples;
  a.alpha = alpha;
  a.weights = weights;
  a.out_rgb = out_rgb;
  a.out_depth = out_depth_raw;
  a.out_normal = out_normal;
  a.out_acc = out_acc;
  a.steps_minmax = steps_minmax;
  a.origins = origins;
  a.bg_density = bg_density;
  a.bg_rgb = bg_rgb;
  a.rgb_merged = rgb_merged;
  if (n_rays == 0) return 0;
  minmax_init_kernel<<<1, 1, 0, s>>>(steps_minmax);
  const unsigned grid = (unsigned)((n_rays + 3) / 4);
  { ProfScope ps_(PS_RENDER_FWD, s); SDFHIP_DISPATCH_C(n_samples, (neus_render_fwd_kernel<C><<<grid, 256, 0, s>>>(a))); }
  depth_clip_kernel<<<(unsigned)((n_rays + 255) / 256), 256, 0, s>>>(out_depth_raw, steps_minmax, (int)n_rays, out_depth);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// VolSDF's compositing (models/volsdf.py:62-79) as one launch per direction: the density-input sibling of neus_render
static void fill_volsdf_render(VolsdfRenderArgs* a, const float* sdf, const float* grad, const float* rgb, const float* starts, const float* ends,
                               const float* beta, const float* background, int64_t n_rays, int32_t n_samples) {
  memset(a, 0, sizeof(*a));
  a->sdf = sdf;
  a->grad = grad;
  a->rgb = rgb;
  a->starts = starts;
  a->ends = ends;
  a->beta = beta;
  a->bg = background;
  a->N = (int)n_rays;
  a->S = n_samples;
}
extern "C" int sdfhip_volsdf_render_forward(const float* sdf, const float* grad, const float* rgb, const float* starts, const float* ends,
                                            const float* beta, const float* background, int64_t n_rays, int32_t n_samples, float* density,
                                            float* weights, float* out_rgb, float* out_depth_raw, float* out_depth, float* out_normal,
                                            float* out_acc, float* bg_trans, float* steps_minmax, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(sdf && grad && rgb && starts && ends && beta && density && weights && out_rgb && out_depth_raw && out_depth && out_normal &&
                     out_acc && bg_trans && steps_minmax, "volsdf_render_forward: null argument");
  SDFHIP_REQUIRE(n_samples >= 1 && n_samples <= 64 * kMaxPerLane, "volsdf_render: n_samples %d unsupported", n_samples);
  if (n_rays == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  VolsdfRenderArgs a;
  fill_volsdf_render(&a, sdf, grad, rgb, starts, ends, beta, background, n_rays, n_samples);
  a.density = density;
  a.weights = weights;
  a.out_rgb = out_rgb;
  a.out_depth = out_depth_raw;
  a.out_normal = out_normal;
  a.out_acc = out_acc;
  a.bg_trans = bg_trans;
  a.steps_minmax = steps_minmax;
  minmax_init_kernel<<<1, 1, 0, s>>>(steps_minmax);
  const unsigned grid = (unsigned)((n_rays + 3) / 4);
  { ProfScope ps_(PS_RENDER_FWD, s); SDFHIP_DISPATCH_C(n_samples, (volsdf_render_fwd_kernel<C><<<grid, 256, 0, s>>>(a))); }
  depth_clip_kernel<<<(unsigned)((n_rays + 255) / 256), 256, 0, s>>>(out_depth_raw, steps_minmax, (int)n_rays, out_depth);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_volsdf_render_backward(const float* sdf, const float* grad, const float* rgb, const float* starts, const float* ends,
                                             const float* beta, const float* background, int64_t n_rays, int32_t n_samples,
                                             const float* density, const float* weights, const float* out_depth_raw, const float* out_acc,
                                             const float* bg_trans, const float* steps_minmax, const float* rgb_bar, const float* depth_bar,
                                             const float* normal_bar, const float* acc_bar, const float* weights_bar,
                                             const float* bg_trans_bar, float* sdf_bar, float* grad_bar, float* rgbs_bar, float* beta_bar,
                                             sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(sdf && grad && rgb && starts && ends && beta && density && weights && out_depth_raw && out_acc && bg_trans && steps_minmax &&
                     sdf_bar && grad_bar && rgbs_bar, "volsdf_render_backward: null argument");
  SDFHIP_REQUIRE(n_samples >= 1 && n_samples <= 64 * kMaxPerLane, "volsdf_render: n_samples %d unsupported", n_samples);
  if (n_rays == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  VolsdfRenderArgs a;
  fill_volsdf_render(&a, sdf, grad, rgb, starts, ends, beta, background, n_rays, n_samples);
  a.density = const_cast<float*>(density);
  a.weights = const_cast<float*>(weights);
  a.out_depth = const_cast<float*>(out_depth_raw);
  a.out_acc = const_cast<float*>(out_acc);
  a.bg_trans = const_cast<float*>(bg_trans);
  a.steps_minmax = const_cast<float*>(steps_minmax);
  a.rgbbar = rgb_bar;
  a.depthbar = depth_bar;
  a.normalbar = normal_bar;
  a.accbar = acc_bar;
  a.weightsbar = weights_bar;
  a.bgtransbar = bg_trans_bar;
  a.sdfbar = sdf_bar;
  a.gradbar = grad_bar;
  a.rgbsbar = rgbs_bar;
  a.betabar = beta_bar;
  const unsigned grid = (unsigned)((n_rays + 3) / 4);
  { ProfScope ps_(PS_RENDER_BWD, s); SDFHIP_DISPATCH_C(n_samples, (volsdf_render_bwd_kernel<C><<<grid, 256, 0, s>>>(a))); }
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

static int neus_render_backward_impl(const float* sdf, const float* grad, const float* rgb, const float* dirs, const float* starts,
                                     const float* ends, const float* variance, const float* background, float cos_anneal,
                                     int64_t n_rays, int32_t n_samples, const float* alpha, const float* weights,
                                     const float* out_depth_raw, const float* out_acc, const float* steps_minmax,
                                     const float* rgb_bar, const float* depth_bar, const float* normal_bar, const float* acc_bar,
                                     const float* weights_bar, float* sdf_bar, float* grad_bar, float* rgbs_bar,
                                     float* variance_bar, const float* origins, const float* bg_density, const float* bg_rgb,
                                     float* bg_density_bar, float* bg_rgb_bar, sdfhip_stream_t stream);
extern "C" int sdfhip_neus_render_backward(const float* sdf, const float* grad, const float* rgb, const float* dirs, const float* starts,
                                           const float* ends, const float* variance, const float* background, float cos_anneal,
                                           int64_t n_rays, int32_t n_samples, const float* alpha, const float* weights,
                                           const float* out_depth_raw, const float* out_acc, const float* steps_minmax,
                                           const float* rgb_bar, const float* depth_bar, const float* normal_bar, const float* acc_bar,
                                           const float* weights_bar, float* sdf_bar, float* grad_bar, float* rgbs_bar,
                                           float* variance_bar, sdfhip_stream_t stream) {
  return neus_render_backward_impl(sdf, grad, rgb, dirs, starts, ends, variance, background, cos_anneal, n_rays, n_samples, alpha, weights,
                                   out_depth_raw, out_acc, steps_minmax, rgb_bar, depth_bar, normal_bar, acc_bar, weights_bar, sdf_bar,
                                   grad_bar, rgbs_bar, variance_bar, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}
extern "C" int sdfhip_neus_render_bg_backward(const float* sdf, const float* grad, const float* rgb, const float* dirs, const float* starts,
                                              const float* ends, const float* variance, const float* background, float cos_anneal,
                                              int64_t n_rays, int32_t n_samples, const float* origins, const float* bg_density,
                                              const float* bg_rgb, const float* alpha, const float* weights, const float* out_depth_raw,
                                              const float* out_acc, const float* steps_minmax, const float* rgb_bar, const float* depth_bar,
                                              const float* normal_bar, const float* acc_bar, const float* weights_bar, float* sdf_bar,
                                              float* grad_bar, float* rgbs_bar, float* variance_bar, float* bg_density_bar,
                                              float* bg_rgb_bar, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(origins && bg_density && bg_rgb && bg_density_bar && bg_rgb_bar, "neus_render_bg_backward: null argument");
  return neus_render_backward_impl(sdf, grad, rgb, dirs, starts, ends, variance, background, cos_anneal, n_rays, n_samples, alpha, weights,
                                   out_depth_raw, out_acc, steps_minmax, rgb_bar, depth_bar, normal_bar, acc_bar, weights_bar, sdf_bar,
                                   grad_bar, rgbs_bar, variance_bar, origins, bg_density, bg_rgb, bg_density_bar, bg_rgb_bar, stream);
}
static int neus_render_backward_impl(const float* sdf, const float* grad, const float* rgb, const float* dirs, const float* starts,
                                     const float* ends, const float* variance, const float* background, float cos_anneal,
                                     int64_t n_rays, int32_t n_samples, const float* alpha, const float* weights,
                                     const float* out_depth_raw, const float* out_acc, const float* steps_minmax,
                                     const float* rgb_bar, const float* depth_bar, const float* normal_bar, const float* acc_bar,
                                     const float* weights_bar, float* sdf_bar, float* grad_bar, float* rgbs_bar,
                                     float* variance_bar, const float* origins, const float* bg_density, const float* bg_rgb,
                                     float* bg_density_bar, float* bg_rgb_bar, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(sdf && grad && rgb && dirs && starts && ends && variance && alpha && weights && out_depth_raw && out_acc && steps_minmax &&
                     sdf_bar && grad_bar && rgbs_bar,
                 "neus_render_backward: null argument");
  SDFHIP_REQUIRE(n_samples >= 1 && n_samples <= 64 * kMaxPerLane, "neus_render: n_samples %d unsupported", n_samples);
  NeusRenderArgs a;
  memset(&a, 0, sizeof(a));
  a.sdf = sdf;
  a.grad = grad;
  a.rgb = rgb;
  a.dirs = dirs;
  a.starts = starts;
  a.ends = ends;
  a.variance = variance;
  a.bg = background;
  a.cos_anneal = cos_anneal;
  a.N = (int)n_rays;
  a.S = n_samples;
  a.alpha = const_cast<float*>(alpha);
  a.weights = const_cast<float*>(weights);
  a.out_depth = const_cast<float*>(out_depth_raw);
  a.out_acc = const_cast<float*>(out_acc);
  a.steps_minmax = const_cast<float*>(steps_minmax);
  a.rgbbar = rgb_bar;
  a.depthbar = depth_bar;
  a.normalbar = normal_bar;
  a.accbar = acc_bar;
  a.weightsbar = weights_bar;
  a.sdfbar = sdf_bar;
  a.gradbar = grad_bar;
  a.rgbsbar = rgbs_bar;
  a.variancebar = variance_bar;
  a.origins = origins;
  a.bg_density = bg_density;
  a.bg_rgb = bg_rgb;
  a.bg_density_bar = bg_density_bar;
  a.bg_rgb_bar = bg_rgb_bar;
  if (n_rays == 0) return 0;
  const unsigned grid = (unsigned)((n_rays + 3) / 4);
  { ProfScope ps_(PS_RENDER_BWD, (hipStream_t)stream); SDFHIP_DISPATCH_C(n_samples, (neus_render_bwd_kernel<C><<<grid, 256, 0, (hipStream_t)stream>>>(a))); }
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}


// ------------------------------------------------------------------------------------------------ ref-nerf colour combination (refnerf_kernels.h)
static constexpr int kRefChunk = 1024;
extern "C" int64_t sdfhip_refnerf_workspace_size(int64_t n_points, int32_t geo_feat_dim) {
  const int64_t nb = (n_points + kRefChunk - 1) / kRefChunk;
  return (int64_t)sizeof(float) * (n_points * 8 + nb * 6 * (geo_feat_dim + 1)) + 512;
}
extern "C" int sdfhip_refnerf_forward(const float* s_rgb, const float* feat, const float* w_d, const float* b_d, const float* w_t,
                                      const float* b_t, int64_t n_points, int32_t geo_feat_dim, float rgb_padding, float* rgb,
                                      sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(s_rgb && feat && w_d && b_d && rgb && (w_t == nullptr) == (b_t == nullptr), "refnerf_forward: null argument");
  SDFHIP_REQUIRE(geo_feat_dim >= 1 && geo_feat_dim <= 1024 && n_points >= 0, "refnerf_forward: bad shape");
  if (n_points == 0) return 0;
  RefCombineArgs a;
  memset(&a, 0, sizeof(a));
  a.h = RefHeads{w_d, b_d, w_t, b_t};
  a.s_rgb = s_rgb;
  a.feat = feat;
  a.n_points = n_points;
  a.gf = geo_feat_dim;
  a.rgb_padding = rgb_padding;
  a.rgb = rgb;
  refnerf_fwd_kernel<<<(unsigned)std::min<int64_t>((n_points + 3) / 4, 8192), 256, 0, (hipStream_t)stream>>>(a);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int sdfhip_refnerf_backward(const float* s_rgb, const float* feat, const float* w_d, const float* b_d, const float* w_t,
                                       const float* b_t, int64_t n_points, int32_t geo_feat_dim, float rgb_padding, const float* rgb_bar,
                                       void* workspace, float* s_bar, float* feat_bar, float* w_d_bar, float* b_d_bar, float* w_t_bar,
                                       float* b_t_bar, sdfhip_stream_t stream) {
  SDFHIP_REQUIRE(s_rgb && feat && w_d && b_d && rgb_bar && workspace && s_bar && feat_bar && w_d_bar && b_d_bar &&
                     (w_t == nullptr) == (b_t == nullptr) && (w_t == nullptr) == (w_t_bar == nullptr) && (w_t == nullptr) == (b_t_bar == nullptr),
                 "refnerf_backward: null argument");
  SDFHIP_REQUIRE(geo_feat_dim >= 1 && geo_feat_dim <= 1024 && n_points >= 0, "refnerf_backward: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int64_t nb = (n_points + kRefChunk - 1) / kRefChunk;
  float* delta = (float*)workspace;
  float* partial = delta + ((n_points * 8 + 63) / 64 * 64);
  if (n_points > 0) {
    RefCombineBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.h = RefHeads{w_d, b_d, w_t, b_t};
    a.s_rgb = s_rgb;
    a.feat = feat;
    a.rgb_bar = rgb_bar;
    a.n_points = n_points;
    a.gf = geo_feat_dim;
    a.rgb_padding = rgb_padding;
    a.s_bar = s_bar;
    a.feat_bar = feat_bar;
    a.delta = delta;
    refnerf_bwd_kernel<<<(unsigned)std::min<int64_t>((n_points + 3) / 4, 8192), 256, 0, s>>>(a);
    RefWgradArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.delta = delta;
    wa.feat = feat;
    wa.n_points = n_points;
    wa.gf = geo_feat_dim;
    wa.chunk = kRefChunk;
    wa.partial = partial;
    refnerf_wgrad_kernel<<<(unsigned)nb, (unsigned)((geo_feat_dim + 63) / 64 * 64), 0, s>>>(wa);
  }
  const int total = 6 * (geo_feat_dim + 1);
  refnerf_wreduce_kernel<<<(total + 255) / 256, 256, 0, s>>>(partial, (int)nb, geo_feat_dim, w_d_bar, b_d_bar, w_t_bar, b_t_bar);
  SDFHIP_CHECK_HIP(hipGetLastError());
  return 0;
}
