/* EXPERIMENTAL entry points of libavc_ring.so -- NOT part of the product ABI (include/avc.h) and not in libavc.so.
 *
 * libavc_ring.so = libavc.so + csrc/avc_bwd_ring.hip, built by `python -m avatarclip_amd.build --ring` (or AVC_WITH_RING=1).  The
 * role-specialised backward hands the abar tiles of the middle SDF layers to accumulator-owning workgroups through an L2-resident ring
 * instead of through the G region; round 4 built it, proved it equal to the panel path (tests/ring_cases.py) and measured it SLOWER
 * with counters (profiles/r04_ring_handoff.md: every handed-off byte still reaches HBM; pair 83.9 vs 80.8 ms).  It stays in the tree as
 * the parity-tested record of that negative result: AVC_LIB_NAME=libavc_ring.so AVC_BWD_RING=1 selects it (scripts/ring_bench.py). */
#ifndef AVC_RING_H
#define AVC_RING_H
#ifdef __cplusplus
extern "C" {
#endif

/* Role-specialised variant of avc_render_points_bwd (same reference lines: main.py:537 through fields.py:96-107): one persistent
 * launch of `grid` workgroups (one per CU) in which, per XCD, ntypes * cpt CONSUMER workgroups each own the fp32 accumulators of
 * one weight-gradient product abar_m (x) h_in of a middle SDF layer (ntypes = avc_bwd_ring_types(net) products, cpt instances
 * each) and every other workgroup is a PRODUCER running the sweeps of avc_render_points_bwd.  The abar tiles of those layers do
 * not go to gpanels: a producer hands each workgroup iteration's tiles to a consumer of its own XCD through a ring of `nslots`
 * (<= 16) slots per (XCD, product) that stays in the XCD's write-back L2 (`ring`, avc_bwd_ring_payload_bytes bytes), the consumer
 * contracts them with the forward-type tiles pb_tiles[type] .. + HT - 1 of fpanels (host array).  All other gradient-type tiles are
 * written to gpanels as before; the caller runs avc_weight_grad_all on the remaining pairs.  Outputs: partial[type][8 cpt]
 * [HT * HT * 1024] and bias_partial[type][8 cpt][HT * 32] (layouts of avc_weight_grad_all's partial / bias_partial for an HT x HT
 * pair; rows of consumers that never ran stay untouched: zero them first), summed by the caller.  ctl = avc_bwd_ring_ctl_bytes()
 * bytes of control words, zeroed by this call; afterwards (u32 index) ctl[64] = number of spin time-outs (0 = ok; otherwise the
 * results are invalid), ctl[65] = first failing site, and the u64 counters at byte 384: [0] producer ticks (10 ns) spent getting
 * a slot, [1] in hand-offs altogether, [2] producer polls, [3] consumer ticks waiting for units, [4] units contracted, [5]
 * consumer workgroups, [6] producer iterations, [7] producer workgroups. */
long avc_bwd_ring_ctl_bytes(void);
long avc_bwd_ring_payload_bytes(int net, int ntypes, int nslots);
int avc_bwd_ring_types(int net);
int avc_render_points_bwd_ring(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z,
                               int S, int ldz, float sample_dist, long npts, const void* wbf16, const float* tab,
                               const int* offs /* host */, const float* d_sdf, const float* d_normal, const float* d_rgb,
                               const float* rgb_fwd, const void* fpanels, void* gpanels, const void* masks,
                               float* colsum /* [8 grid][avc_bwd_colsum_floats(net)], zeroed by the caller */, void* ctl,
                               void* ring, float* partial, float* bias_partial, const int* pb_tiles /* host */, int ntypes,
                               int cpt, int nslots, int grid, void* stream);

#ifdef __cplusplus
}
#endif
#endif
