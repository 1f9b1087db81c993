// tuning.hip -- the one place the library reads its environment (tuning.h).
#include <cstdio>
#include <cstdlib>
#include "tuning.h"

namespace tf {
namespace {
bool flag(const char* name) { return getenv(name) != nullptr; }
long num(const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; }
Tuning parse() {
  Tuning t;
  t.ew_blocks = (int)num("TINYFACES_EW_BLOCKS", 1024);
  t.ew_blocks_small = (int)num("TINYFACES_EW_BLOCKS_SMALL", 768);
  t.ew_small_mb = (int)num("TINYFACES_EW_SMALL_MB", 40);
  t.comm_fail_init = flag("TINYFACES_COMM_FAIL_INIT");
  t.comm_fail_bucket = (int)num("TINYFACES_COMM_FAIL_BUCKET", -1);
  t.pws_off = flag("TINYFACES_PWS_OFF");
  t.pwx_fwd = flag("TINYFACES_PWX_FWD");
  t.t12_shortk_off = flag("TINYFACES_T12_SHORTK_OFF");
  t.t46_shortk_off = flag("TINYFACES_T46_SHORTK_OFF");
  t.shortk_big_tile = (int)num("TINYFACES_SHORTK_BIG_TILE", 46);
  t.t46_handover_min_m = num("TINYFACES_T46_HANDOVER_MIN_M", 16384);
  t.shortk_tile = (int)num("TINYFACES_SHORTK_TILE", 32);
  t.mma32_off = flag("TINYFACES_MMA32_OFF");
  t.conv3h_dbg = (int)num("TINYFACES_CONV3H_DBG", 0);
  t.conv3h_minblocks = (int)num("TINYFACES_CONV3H_MINBLOCKS", 160);
  t.epi_spec_off = flag("TINYFACES_EPI_SPEC_OFF");
  t.conv3h_off = flag("TINYFACES_CONV3H_OFF");
  t.conv3h_mincin = (int)num("TINYFACES_CONV3H_MINCIN", 256);
  t.conv3h_tr6 = num("TINYFACES_CONV3H_TR6", 1) != 0;
  t.pws_sliced = flag("TINYFACES_PWS_SLICED");
  t.stem_direct_off = flag("TINYFACES_STEM_DIRECT_OFF");
  t.wgrad_group = (int)num("TINYFACES_WGRAD_GROUP", 8);
  t.fork_by_record = flag("TINYFACES_FORK_BY_RECORD");
  t.stat_shift_off = flag("TINYFACES_STAT_SHIFT_OFF");
  t.side_prio_low = flag("TINYFACES_SIDE_PRIO_LOW");
  t.unfused_bn = flag("TINYFACES_UNFUSED_BN");
  t.pack_side = flag("TINYFACES_PACK_SIDE");
  t.pack_split_off = flag("TINYFACES_PACK_SPLIT_OFF");
  t.single_stream = flag("TINYFACES_SINGLE_STREAM");
  t.pack_fork_late = flag("TINYFACES_PACK_FORK_LATE");
  t.pack_first_side = flag("TINYFACES_PACK_FIRST_SIDE");
  t.bnf = flag("TINYFACES_BNF");
  t.dbg_skip_wgrad = flag("TINYFACES_DBG_SKIP_WGRAD");
  t.wgrad3_off = flag("TINYFACES_WGRAD3_OFF");
  t.group_stream = flag("TINYFACES_GROUP_STREAM");
  t.stem_wgrad_im2col = flag("TINYFACES_STEM_WGRAD_IM2COL");
  t.grad_memset_full = flag("TINYFACES_GRAD_MEMSET_FULL");
  t.fork_per_block = flag("TINYFACES_FORK_PER_BLOCK");
  t.l3_fork_per_wgrad = flag("TINYFACES_L3_FORK_PER_WGRAD");
  t.dbg_group_refuse = flag("TINYFACES_DBG_GROUP_REFUSE");
  t.wgradg_split = (int)num("TINYFACES_WGRADG_SPLIT", 1);
  t.pwx_all = flag("TINYFACES_PWX_ALL");
  t.pwx_off = flag("TINYFACES_PWX_OFF");
  t.pwx_bwd = flag("TINYFACES_PWX_BWD");
  t.handover_tile = (int)num("TINYFACES_HANDOVER_TILE", 0);
  t.pool_stats_off = flag("TINYFACES_POOL_STATS_OFF");
  t.stem_apply_separate = flag("TINYFACES_STEM_APPLY_SEPARATE");
  t.pool_stats_blocks = (int)num("TINYFACES_POOL_STATS_BLOCKS", 8192);
  t.profile_bracket = flag("TINYFACES_PROFILE_BRACKET");
  t.stem_wgrad_blocks = (int)num("TINYFACES_STEM_WGRAD_BLOCKS", 384);
  t.wgrad3_blocks = (int)num("TINYFACES_WGRAD3_BLOCKS", 128);
  t.wgrad3_atomics = flag("TINYFACES_WGRAD3_ATOMICS");
  t.wgrad3_dbg = (int)num("TINYFACES_WGRAD3_DBG", 0);
  t.wgrad_blocks = (int)num("TINYFACES_WGRAD_BLOCKS", 512);
  t.wgrad_ns = (int)num("TINYFACES_WGRAD_NS", 3);
  t.dma_builtin = flag("TINYFACES_DMA_BUILTIN");
  t.wgradg_fast = (int)num("TINYFACES_WGRADG_FAST", 8);
  t.conv_dbg = (int)num("TF_CONV_DBG", 0);
  t.scatter_dgrad_off = flag("TINYFACES_SCATTER_DGRAD_OFF");
  t.ds_inplace_off = flag("TINYFACES_DS_INPLACE_OFF");
  t.parity_dgrad_off = flag("TINYFACES_PARITY_DGRAD_OFF");
  t.ns2_maxstages = (int)num("TINYFACES_NS2_MAXSTAGES", 16);
  t.ns1_maxstages = (int)num("TINYFACES_NS1_MAXSTAGES", 4);
  // clamps the call sites used to apply themselves
  t.wgrad_group = t.wgrad_group < 0 ? 0 : (t.wgrad_group > 22 ? 22 : t.wgrad_group);
  t.wgradg_split = t.wgradg_split < 1 ? 1 : t.wgradg_split;
  t.wgrad_ns = t.wgrad_ns == 2 ? 2 : 3;
  // modes that make results invalid say so, once
  if (t.conv3h_dbg) fprintf(stderr, "tinyfaces: TINYFACES_CONV3H_DBG=%d -- timing-ablation mode, convolution RESULTS ARE INVALID\n", t.conv3h_dbg);
  if (t.conv_dbg & 15) fprintf(stderr, "tinyfaces: TF_CONV_DBG=%d -- timing-ablation mode, convolution RESULTS ARE INVALID\n", t.conv_dbg);
  else if (t.conv_dbg) fprintf(stderr, "tinyfaces: TF_CONV_DBG=%d -- A/B form of the statistic epilogue (results unchanged)\n", t.conv_dbg);
  if (t.dbg_skip_wgrad) fprintf(stderr, "tinyfaces: TINYFACES_DBG_SKIP_WGRAD -- weight gradients NOT computed, timing only\n");
  if (t.wgrad3_dbg) fprintf(stderr, "tinyfaces: TINYFACES_WGRAD3_DBG=%d -- timing-ablation mode, weight gradients ARE INVALID\n", t.wgrad3_dbg);
  return t;
}
}  // namespace
const Tuning& tuning() {
  static const Tuning t = parse();
  return t;
}
}  // namespace tf
