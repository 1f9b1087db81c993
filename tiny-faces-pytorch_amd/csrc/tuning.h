// tuning.h -- every run-time knob of the library in ONE struct, parsed from the environment ONCE (r6; rounds 1-5 had 64 getenv() calls spread
// over the kernel files, each behind its own function-local static).  tf::tuning() returns the parsed values; a knob is an A/B or measurement
// switch named in DESIGN.md / profiles/ -- defaults are what the product runs.  Invalid-result modes announce themselves on stderr once.
#pragma once
namespace tf {
struct Tuning {
  int ew_blocks = 1024;                  // TINYFACES_EW_BLOCKS: block cap of the fused BN passes (r6: 2048 -> 1024, +1.0 % on the step)
  int ew_blocks_small = 768;             // TINYFACES_EW_BLOCKS_SMALL: the cap for tensors below ew_small_mb
  int ew_small_mb = 40;                  // TINYFACES_EW_SMALL_MB
  bool comm_fail_init = false;           // TINYFACES_COMM_FAIL_INIT
  int comm_fail_bucket = -1;             // TINYFACES_COMM_FAIL_BUCKET
  bool pws_off = false;                  // TINYFACES_PWS_OFF
  bool pwx_fwd = false;                  // TINYFACES_PWX_FWD
  bool t12_shortk_off = false;           // TINYFACES_T12_SHORTK_OFF
  bool t46_shortk_off = false;           // TINYFACES_T46_SHORTK_OFF
  int shortk_big_tile = 46;              // TINYFACES_SHORTK_BIG_TILE
  long t46_handover_min_m = 16384;       // TINYFACES_T46_HANDOVER_MIN_M
  int shortk_tile = 32;                  // TINYFACES_SHORTK_TILE
  bool mma32_off = false;                // TINYFACES_MMA32_OFF
  int conv3h_dbg = 0;                    // TINYFACES_CONV3H_DBG
  int conv3h_minblocks = 160;            // TINYFACES_CONV3H_MINBLOCKS
  bool epi_spec_off = false;             // TINYFACES_EPI_SPEC_OFF
  bool conv3h_off = false;               // TINYFACES_CONV3H_OFF
  int conv3h_mincin = 256;               // TINYFACES_CONV3H_MINCIN
  bool conv3h_tr6 = true;                // TINYFACES_CONV3H_TR6 (0: evaluation launches never take the 6-row tile)
  bool pws_sliced = false;               // TINYFACES_PWS_SLICED
  bool stem_direct_off = false;          // TINYFACES_STEM_DIRECT_OFF
  int wgrad_group = 8;                   // TINYFACES_WGRAD_GROUP
  bool fork_by_record = false;           // TINYFACES_FORK_BY_RECORD
  bool stat_shift_off = false;           // TINYFACES_STAT_SHIFT_OFF
  bool side_prio_low = false;            // TINYFACES_SIDE_PRIO_LOW
  bool unfused_bn = false;               // TINYFACES_UNFUSED_BN
  bool pack_side = false;                // TINYFACES_PACK_SIDE
  bool pack_split_off = false;           // TINYFACES_PACK_SPLIT_OFF
  bool single_stream = false;            // TINYFACES_SINGLE_STREAM
  bool pack_fork_late = false;           // TINYFACES_PACK_FORK_LATE
  bool pack_first_side = false;          // TINYFACES_PACK_FIRST_SIDE
  bool bnf = false;                      // TINYFACES_BNF
  bool dbg_skip_wgrad = false;           // TINYFACES_DBG_SKIP_WGRAD
  bool wgrad3_off = false;               // TINYFACES_WGRAD3_OFF
  bool group_stream = false;             // TINYFACES_GROUP_STREAM
  bool stem_wgrad_im2col = false;        // TINYFACES_STEM_WGRAD_IM2COL
  bool grad_memset_full = false;         // TINYFACES_GRAD_MEMSET_FULL
  bool fork_per_block = false;           // TINYFACES_FORK_PER_BLOCK
  bool l3_fork_per_wgrad = false;        // TINYFACES_L3_FORK_PER_WGRAD
  bool dbg_group_refuse = false;         // TINYFACES_DBG_GROUP_REFUSE
  int wgradg_split = 1;                  // TINYFACES_WGRADG_SPLIT
  bool pwx_all = false;                  // TINYFACES_PWX_ALL
  bool pwx_off = false;                  // TINYFACES_PWX_OFF
  bool pwx_bwd = false;                  // TINYFACES_PWX_BWD
  int handover_tile = 0;                 // TINYFACES_HANDOVER_TILE
  bool pool_stats_off = false;           // TINYFACES_POOL_STATS_OFF
  bool stem_apply_separate = false;      // TINYFACES_STEM_APPLY_SEPARATE
  int pool_stats_blocks = 8192;          // TINYFACES_POOL_STATS_BLOCKS
  bool profile_bracket = false;          // TINYFACES_PROFILE_BRACKET
  int stem_wgrad_blocks = 384;           // TINYFACES_STEM_WGRAD_BLOCKS
  int wgrad3_blocks = 128;               // TINYFACES_WGRAD3_BLOCKS (r6: 256 -> 128 pixel slices x tiles: half the partial-tile traffic, +0.4 % on the step)
  bool wgrad3_atomics = false;           // TINYFACES_WGRAD3_ATOMICS
  int wgrad3_dbg = 0;                    // TINYFACES_WGRAD3_DBG
  int wgrad_blocks = 512;                // TINYFACES_WGRAD_BLOCKS
  int wgrad_ns = 3;                      // TINYFACES_WGRAD_NS
  bool dma_builtin = false;              // TINYFACES_DMA_BUILTIN
  int wgradg_fast = 8;                   // TINYFACES_WGRADG_FAST
  int conv_dbg = 0;                      // TF_CONV_DBG
  bool scatter_dgrad_off = false;        // TINYFACES_SCATTER_DGRAD_OFF
  bool ds_inplace_off = false;           // TINYFACES_DS_INPLACE_OFF (1: zero / copy the raster, scatter, hand over with it as the residual: rounds 3-5)
  bool parity_dgrad_off = false;         // TINYFACES_PARITY_DGRAD_OFF
  int ns2_maxstages = 16;                // TINYFACES_NS2_MAXSTAGES
  int ns1_maxstages = 4;                 // TINYFACES_NS1_MAXSTAGES
};
const Tuning& tuning();
}  // namespace tf
