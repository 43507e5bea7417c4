#include "halo_plan.h"

#include <algorithm>
#include <map>

namespace {

struct Table {
    std::vector<HaloBlock> blk;
    // blocks of each rank ordered by local index
    std::map<int, std::vector<int>> by_rank;
    int find(int ig, int jg) const
    {
        for (size_t k = 0; k < blk.size(); ++k) {
            const HaloBlock &b = blk[k];
            if (ig >= b.gi0 && ig < b.gi0 + b.gnx && jg >= b.gj0 && jg < b.gj0 + b.gny) return (int)k;
        }
        return -1;
    }
};

struct Src {
    bool outside = false;   // beyond a closed/open outer boundary: ghost left untouched
    int ig = 0, jg = 0;
    int sign = 1;
};

// Which global cell does the ghost position (ig,jg) of an NE-corner vector field mirror?
Src resolve(const cice_evp_hip_dims &d, int ig, int jg)
{
    Src s;
    const int NX = d.nx_global, NY = d.ny_global;
    if (ig < 1 || ig > NX) {
        if (d.ew_boundary_type == CICE_EVP_BND_CYCLIC) ig = (ig < 1) ? ig + NX : ig - NX;
        else s.outside = true;
    }
    if (jg < 1) {
        if (d.ns_boundary_type == CICE_EVP_BND_CYCLIC) jg += NY;
        else s.outside = true;
    } else if (jg > NY) {
        if (d.ns_boundary_type == CICE_EVP_BND_CYCLIC) jg -= NY;
        else if (d.ns_boundary_type == CICE_EVP_BND_TRIPOLET && !s.outside) {
            // T-fold, NE-corner vector field (ice_boundary.F90:1563-1622 offsets (0, 1), copy-out :1686-1722 with the
            // buffer addresses of :8135-8159): ghost(ig, NY+1) <- - a(NX-ig+1, NY-2)
            ig = NX - ig + 1;
            jg = NY - 2;
            s.sign = -1;
        } else if (d.ns_boundary_type == CICE_EVP_BND_TRIPOLE && !s.outside) {
            // u-fold mirror of an NE-corner vector field (ice_blocks.F90:423-424;
            // copy-out offsets (1,1) and isign = -1, ice_boundary.F90:1555-1556,1632-1633):
            //   ghost(ig, NY+k) <- - a(NX-ig, NY-k)
            const int k = jg - NY;
            ig = NX - ig;
            if (ig < 1) ig += NX;
            jg = NY - k;
            s.sign = -1;
        } else s.outside = true;
    }
    else if (jg == NY && d.ns_boundary_type == CICE_EVP_BND_TRIPOLET && !s.outside) {
        // ... and the top physical row itself (interior cells and their east-west ghost columns) is the image of row
        // NY-1: a(ig, NY) <- - a(NX-ig+1, NY-1); nothing is averaged at this location
        ig = NX - ig + 1;
        jg = NY - 1;
        s.sign = -1;
    }
    s.ig = ig;
    s.jg = jg;
    return s;
}

}  // namespace

bool build_halo_plan(const cice_evp_hip_dims &d, HaloPlan &plan)
{
    plan = HaloPlan();
    plan.nx_block = d.nx_block;
    plan.ny_block = d.ny_block;
    plan.nblocks = d.nblocks;
    if (d.nghost != 1) {
        plan.error = "nghost must be 1 (ice_blocks.F90:47)";
        return false;
    }
    const bool tripole = d.ns_boundary_type == CICE_EVP_BND_TRIPOLE;
    const bool tfold = d.ns_boundary_type == CICE_EVP_BND_TRIPOLET;
    plan.tfold = tfold;
    if ((tripole || tfold) && (d.nx_global % 2 != 0 || d.ew_boundary_type != CICE_EVP_BND_CYCLIC)) {
        plan.error = "tripole needs an even nx_global and a cyclic east-west boundary";
        return false;
    }
    // (tripoleT on several ranks: the images of the top row are INTERIOR cells -- receive lists may name them; the exchange
    // then has to follow the launch that computes them, never ride in it: evp_host_loop.cpp use_riding_exchange / use_overlap)
    const int ng = d.nghost;
    const int nx = d.nx_block, ny = d.ny_block;
    const size_t plane = (size_t)nx * ny;
    const int me = d.rank;

    Table T;
    if (d.gi0 != nullptr && d.nblocks_tot > 0) {
        for (int k = 0; k < d.nblocks_tot; ++k)
            T.blk.push_back({d.gi0[k], d.gj0[k], d.gnx[k], d.gny[k], d.gowner[k], d.glocal[k]});
    } else {
        if (d.nranks != 1) {
            plan.error = "global block table required when nranks > 1";
            return false;
        }
        for (int b = 0; b < d.nblocks; ++b)
            T.blk.push_back({d.iglob0[b], d.jglob0[b], d.ihi[b] - d.ilo[b] + 1,
                             d.jhi[b] - d.jlo[b] + 1, me, b});
    }
    for (size_t k = 0; k < T.blk.size(); ++k)
        if (T.blk[k].owner >= 0) T.by_rank[T.blk[k].owner].push_back((int)k);
    for (auto &kv : T.by_rank)
        std::sort(kv.second.begin(), kv.second.end(),
                  [&](int a, int b) { return T.blk[a].local < T.blk[b].local; });

    // consistency of the local description with the table
    {
        auto it = T.by_rank.find(me);
        const size_t nloc = (it == T.by_rank.end()) ? 0 : it->second.size();
        if ((int)nloc != d.nblocks) {
            plan.error = "global block table disagrees with nblocks of this rank";
            return false;
        }
        for (int b = 0; b < d.nblocks; ++b) {
            const HaloBlock &B = T.blk[it->second[b]];
            if (d.ilo[b] != ng + 1 || d.jlo[b] != ng + 1 || B.local != b || B.gi0 != d.iglob0[b] ||
                B.gj0 != d.jglob0[b] || B.gnx != d.ihi[b] - d.ilo[b] + 1 ||
                B.gny != d.jhi[b] - d.jlo[b] + 1 || d.ihi[b] + ng > nx || d.jhi[b] + ng > ny) {
                plan.error = "local block geometry inconsistent with the global block table";
                return false;
            }
        }
    }

    if (tfold) {        // owners of the blocks that hold rows NY-2 .. NY (the C grid's fold step reads all three)
        bool mine = false, others = false;
        for (const HaloBlock &B : T.blk) {
            if (B.owner < 0 || B.gj0 + B.gny - 1 < d.ny_global - 2) continue;
            (B.owner == me ? mine : others) = true;
        }
        plan.fold_rows = !mine ? 0 : (others ? 2 : 1);
    }
    if (tripole) {      // owners of the blocks that hold rows NY-1 / NY
        bool mine = false, others = false;
        for (const HaloBlock &B : T.blk) {
            if (B.owner < 0 || B.gj0 + B.gny - 1 < d.ny_global - 1) continue;
            (B.owner == me ? mine : others) = true;
        }
        plan.fold_rows = !mine ? 0 : (others ? 2 : 1);
        int first = -1;
        for (const HaloBlock &B : T.blk) {
            if (B.owner < 0 || B.gj0 + B.gny - 1 != d.ny_global) continue;
            if (first < 0) first = B.owner;
            else if (B.owner != first) plan.fold_split = true;
        }
    }

    std::map<int, HaloPeer> peers;
    struct GhostSeam { int R; int32_t dst; int sig; int sign; };
    std::vector<GhostSeam> ghost_seam;           // ghost cells (of any rank) that mirror a seam-row cell, canonical order

    // Enumerate the ghost cells of every rank in one canonical order (local
    // block index, then j, then i).  The receiver keeps entries whose source it
    // does not own in recv lists; the owner of the source, running the very same
    // enumeration, appends the matching cell to its send list -- so both lists
    // have identical order without any set-up communication.
    for (const auto &kv : T.by_rank) {
        const int R = kv.first;
        for (int kb : kv.second) {
            const HaloBlock &B = T.blk[kb];
            const int ilo = ng + 1, jlo = ng + 1, ihi = ng + B.gnx, jhi = ng + B.gny;
            for (int j = jlo - ng; j <= jhi + ng; ++j)
                for (int i = ilo - ng; i <= ihi + ng; ++i) {
                    const bool interior = i >= ilo && i <= ihi && j >= jlo && j <= jhi;
                    // (tripoleT: the top physical row is a destination of the halo update as well)
                    if (interior && !(tfold && B.gj0 + (j - jlo) == d.ny_global)) continue;
                    const Src s = resolve(d, B.gi0 + (i - ilo), B.gj0 + (j - jlo));
                    if (s.outside) continue;
                    const int32_t dst = (int32_t)((size_t)B.local * plane + (size_t)(j - 1) * nx + (i - 1));
                    const int ks = T.find(s.ig, s.jg);
                    if (ks < 0 || T.blk[ks].owner < 0) {
                        // eliminated land block: reference fills with 0 (srcBlock == 0)
                        if (R == me) {
                            plan.local_dst.push_back(dst);
                            plan.local_src.push_back(-1);
                            plan.local_sign.push_back(1);
                        }
                        continue;
                    }
                    const HaloBlock &S = T.blk[ks];
                    const int32_t src = (int32_t)((size_t)S.local * plane +
                                                  (size_t)(ng + (s.jg - S.gj0)) * nx + (ng + (s.ig - S.gi0)));
                    const bool src_on_seam = tripole && s.jg == d.ny_global;
                    if (src_on_seam) {
                        // finalised after the exchange from RAW pair values (fin lists below); the plain copy only
                        // stays in the local lists (late_*: single-rank form of the same step)
                        ghost_seam.push_back({R, dst, s.ig, s.sign});
                        if (S.owner != R) {
                            // (on-chip kernel: the owner's final value as a record of its own, see halo_plan.h)
                            if (S.owner == me) {
                                HaloPeer &p = peers[R];
                                p.rank = R;
                                p.fimg_src.push_back(src); p.fimg_dst.push_back(dst); p.fimg_sign.push_back((int8_t)s.sign);
                            } else if (R == me) {
                                HaloPeer &p = peers[S.owner];
                                p.rank = S.owner;
                                p.fimg_recv_dst.push_back(dst); p.fimg_recv_col.push_back(s.ig); p.fimg_recv_sign.push_back((int8_t)s.sign);
                            }
                            continue;
                        }
                    }
                    if (S.owner != R && s.sign < 0) plan.any_fold_exchange = true;
                    if (R == me) {
                        if (S.owner == me) {
                            plan.local_dst.push_back(dst);
                            plan.local_src.push_back(src);
                            plan.local_sign.push_back((int8_t)s.sign);
                            if (src_on_seam) {
                                plan.late_dst.push_back(dst);
                                plan.late_src.push_back(src);
                                plan.late_sign.push_back((int8_t)s.sign);
                            }
                        } else {
                            HaloPeer &p = peers[S.owner];
                            p.rank = S.owner;
                            p.recv_dst.push_back(dst);
                            p.recv_sign.push_back((int8_t)s.sign);
                            p.recv_gid.push_back((int32_t)((s.ig - 1) + (size_t)d.nx_global * (s.jg - 1)));
                        }
                    } else if (S.owner == me) {
                        HaloPeer &p = peers[R];
                        p.rank = R;
                        p.send_src.push_back(src);
                        p.send_dst.push_back(dst);
                        p.send_sign.push_back((int8_t)s.sign);
                    }
                }
        }
    }
    for (auto &kv : peers) { kv.second.n_ghost_send = (int)kv.second.send_src.size(); kv.second.n_ghost_recv = (int)kv.second.recv_dst.size(); }
    for (auto &kv : peers) plan.peers.push_back(kv.second);      // (the tripole section may append staging entries and rebuilds this)

    // cell-centre fields: ghosts of this rank's blocks, same enumeration
    {
        const int NX = d.nx_global, NY = d.ny_global;
        auto it = T.by_rank.find(me);
        if (it != T.by_rank.end())
            for (int kb : it->second) {
                const HaloBlock &B = T.blk[kb];
                const int ilo = ng + 1, jlo = ng + 1, ihi = ng + B.gnx, jhi = ng + B.gny;
                for (int j = jlo - ng; j <= jhi + ng; ++j)
                    for (int i = ilo - ng; i <= ihi + ng; ++i) {
                        if (i >= ilo && i <= ihi && j >= jlo && j <= jhi) continue;
                        int ig = B.gi0 + (i - ilo), jg = B.gj0 + (j - jlo), sign = 1;
                        bool outside = false;
                        if (ig < 1 || ig > NX) {
                            if (d.ew_boundary_type == CICE_EVP_BND_CYCLIC) ig = (ig < 1) ? ig + NX : ig - NX;
                            else outside = true;
                        }
                        if (jg < 1) {
                            if (d.ns_boundary_type == CICE_EVP_BND_CYCLIC) jg += NY;
                            else outside = true;
                        } else if (jg > NY) {
                            if (d.ns_boundary_type == CICE_EVP_BND_CYCLIC) jg -= NY;
                            else if (tripole && !outside) {
                                const int k = jg - NY;
                                ig = NX - ig + 1;
                                jg = NY - k + 1;
                                sign = -1;
                            } else outside = true;      // (tripoleT: no centre lists -- the preparation stays with the host)
                        }
                        if (outside) continue;
                        const int32_t dst = (int32_t)((size_t)B.local * plane + (size_t)(j - 1) * nx + (i - 1));
                        const int ks = T.find(ig, jg);
                        if (ks < 0 || T.blk[ks].owner < 0) {       // eliminated land block: 0
                            plan.center_dst.push_back(dst);
                            plan.center_src.push_back(-1);
                            plan.center_vsign.push_back(1);
                            continue;
                        }
                        const HaloBlock &Sb = T.blk[ks];
                        if (Sb.owner != me) {
                            plan.center_remote = true;
                            if (sign < 0) {
                                plan.center_fold_remote = true;
                                plan.center_foldr_dst.push_back(dst);
                            }
                            continue;
                        }
                        plan.center_dst.push_back(dst);
                        plan.center_src.push_back((int32_t)((size_t)Sb.local * plane +
                                                            (size_t)(ng + (jg - Sb.gj0)) * nx + (ng + (ig - Sb.gi0))));
                        plan.center_vsign.push_back((int8_t)sign);
                    }
            }
    }

    if (tfold) {       // cell-centre fields on the T-fold: rows NY (on the fold) and NY+1 of this rank's blocks, ghost columns included
        const int NX = d.nx_global, NY = d.ny_global;
        auto cell_of = [&](int ig, int jg) -> int32_t {
            const int k = T.find(ig, jg);
            if (k < 0 || T.blk[k].owner != me) { plan.center_tf_remote = true; return -1; }
            const HaloBlock &B = T.blk[k];
            return (int32_t)((size_t)B.local * plane + (size_t)(ng + (jg - B.gj0)) * nx + (ng + (ig - B.gi0)));
        };
        auto it = T.by_rank.find(me);
        if (it != T.by_rank.end())
            for (int kb : it->second) {
                const HaloBlock &B = T.blk[kb];
                const int ilo = ng + 1, jlo = ng + 1, ihi = ng + B.gnx, jhi = ng + B.gny;
                for (int j = jlo - ng; j <= jhi + ng; ++j) {
                    const int jg = B.gj0 + (j - jlo);
                    if (jg != NY && jg != NY + 1) continue;
                    for (int i = ilo - ng; i <= ihi + ng; ++i) {
                        int ig = B.gi0 + (i - ilo);
                        if (ig < 1) ig += NX;
                        if (ig > NX) ig -= NX;
                        int m = NX - ig + 2;
                        if (m > NX) m -= NX;
                        int32_t a, b = -1;
                        uint8_t flip = 1;
                        if (jg == NY + 1) a = cell_of(m, NY - 1);
                        else if (ig == 1 || ig == NX / 2 + 1) a = cell_of(ig, NY);
                        else if (ig <= NX / 2) { a = cell_of(ig, NY); b = cell_of(m, NY); flip = 0; }
                        else { a = cell_of(m, NY); b = cell_of(ig, NY); }
                        plan.center_tf_dst.push_back((int32_t)((size_t)B.local * plane + (size_t)(j - 1) * nx + (i - 1)));
                        plan.center_tf_a.push_back(a);
                        plan.center_tf_b.push_back(b);
                        plan.center_tf_flip.push_back(flip);
                    }
                }
            }
    }

    if (tripole) {
        const int NX = d.nx_global, NY = d.ny_global;
        auto offset_of = [&](int ig, int jg, int &owner) -> int32_t {
            const int k = T.find(ig, jg);
            if (k < 0 || T.blk[k].owner < 0) { owner = -1; return -1; }
            const HaloBlock &B = T.blk[k];
            owner = B.owner;
            return (int32_t)((size_t)B.local * plane + (size_t)(ng + (jg - B.gj0)) * nx + (ng + (ig - B.gi0)));
        };
        for (int ig = 1; ig <= NX; ++ig) {       // pairs with both halves on this rank (single-rank form; on-chip kernel)
            int oa = -1, ob = -1;
            const int32_t a = offset_of(ig, NY, oa);
            if (ig == NX / 2 || ig == NX) {
                if (oa == me) plan.seam_pole.push_back(a);
                continue;
            }
            if (ig > NX / 2 - 1) continue;      // pairs are enumerated from their low index
            const int32_t b = offset_of(NX - ig, NY, ob);
            if (oa != me || ob != me) continue;
            plan.seam_a.push_back(a);
            plan.seam_b.push_back(b);
        }
        // General form: what every rank R must finalise, and which raw seam values of other ranks it needs for
        // that.  Every rank runs the same enumeration for every R, so that a needed value appears at the same
        // position of R's recv list and of its owner's send list.
        for (const auto &kv : T.by_rank) {
            const int R = kv.first;
            const int32_t nR = (int32_t)(plane * kv.second.size());
            std::map<int, int32_t> slot_of;          // global column of a remote seam cell -> staging slot of R
            int32_t next_slot = 0;
            auto ref = [&](int ig) -> int32_t {     // offset, at R, of the RAW value of seam cell (ig, NY); -1: eliminated
                int ow = -1;
                const int32_t off = offset_of(ig, NY, ow);
                if (ow < 0) return -1;
                if (ow == R) return off;
                auto it = slot_of.find(ig);
                if (it != slot_of.end()) return it->second;
                const int32_t slot = nR + next_slot++;
                slot_of[ig] = slot;
                plan.any_fold_exchange = true;
                if (R == me) {
                    HaloPeer &p = peers[ow];
                    p.rank = ow;
                    p.recv_dst.push_back(slot);
                    p.recv_sign.push_back(1);
                    p.recv_gid.push_back((int32_t)((ig - 1) + (size_t)NX * (NY - 1)));
                } else if (ow == me) {
                    HaloPeer &p = peers[R];
                    p.rank = R;
                    p.send_src.push_back(off);
                    p.send_dst.push_back(slot);
                    p.send_sign.push_back(1);
                }
                return slot;
            };
            auto finalise = [&](int32_t dst, int sig, int sign) {   // dst takes sign * (final value of seam cell sig)
                int32_t fa, fb = -1;
                int coef = sign;
                if (sig == NX / 2 || sig == NX) {
                    fa = ref(sig);
                    coef = -sign;                                   // pole: x <- -x
                } else {
                    const int lo = std::min(sig, NX - sig), hi = NX - lo;
                    const int32_t ra = ref(lo), rb = ref(hi);
                    if (ra < 0 || rb < 0) { fa = ref(sig); }        // partner eliminated: nothing to average
                    else { fa = ra; fb = rb; if (sig == hi) coef = -sign; }
                }
                if (R == me && fa >= 0) {
                    plan.fin_dst.push_back(dst);
                    plan.fin_a.push_back(fa);
                    plan.fin_b.push_back(fb);
                    plan.fin_coef.push_back((int8_t)coef);
                }
            };
            for (int ig = 1; ig <= NX; ++ig) {       // R's own seam-row cells
                int ow = -1;
                const int32_t off = offset_of(ig, NY, ow);
                if (ow == R) finalise(off, ig, 1);
            }
            for (const GhostSeam &g : ghost_seam)    // R's ghost images of seam-row cells
                if (g.R == R) {
                    finalise(g.dst, g.sig, g.sign);
                    if (R == me && g.sign > 0) {     // an east-west image in row NY itself: for centre fields, the raw value
                        const int32_t slot = ref(g.sig);
                        if (slot >= nR) { plan.center_seam_dst.push_back(g.dst); plan.center_seam_slot.push_back(slot); }
                    }
                }
            if (R == me) plan.tail = next_slot;
        }
        plan.peers.clear();
        for (auto &kv2 : peers) plan.peers.push_back(kv2.second);
        // stress symmetrisation lists (cell-centre fold: partner column NX-ig+1)
        auto it = T.by_rank.find(me);
        if (it != T.by_rank.end())
            for (int kb : it->second) {
                const HaloBlock &B = T.blk[kb];
                if (B.gj0 + B.gny - 1 != NY) continue;            // not a top-row block
                const int j = ng + B.gny + 1;                     // local ghost row = global NY+1
                for (int i = 1; i <= B.gnx + 2 * ng; ++i) {
                    int ig = B.gi0 + (i - (ng + 1));
                    if (ig < 1) ig += NX;
                    if (ig > NX) ig -= NX;
                    int owner = -1;
                    const int32_t src = offset_of(NX - ig + 1, NY, owner);
                    if (owner >= 0 && owner != me) {
                        // partner on another rank: through the exchange of a shifted copy (halo_plan.h)
                        plan.stress_remote = true;
                        plan.stress_foldr_dst.push_back((int32_t)((size_t)B.local * plane + (size_t)(j - 1) * nx + (i - 1)));
                        continue;
                    }
                    plan.stress_dst.push_back((int32_t)((size_t)B.local * plane + (size_t)(j - 1) * nx + (i - 1)));
                    plan.stress_src.push_back(owner < 0 ? -1 : src);
                }
            }
    }
    if (tfold) {         // stress symmetrisation lists of the T-fold (halo_plan.h)
        const int NX = d.nx_global, NY = d.ny_global;
        auto cell_of = [&](int ig, int &owner) -> int32_t {
            const int k = T.find(ig, NY);
            if (k < 0 || T.blk[k].owner < 0) { owner = -1; return -1; }
            const HaloBlock &B = T.blk[k];
            owner = B.owner;
            return (int32_t)((size_t)B.local * plane + (size_t)(ng + (NY - B.gj0)) * nx + (ng + (ig - B.gi0)));
        };
        auto cell_below = [&](int ig, int &owner) -> int32_t {            // (ig, NY-1)
            const int k = T.find(ig, NY - 1);
            if (k < 0 || T.blk[k].owner < 0) { owner = -1; return -1; }
            const HaloBlock &B = T.blk[k];
            owner = B.owner;
            return (int32_t)((size_t)B.local * plane + (size_t)(ng + (NY - 1 - B.gj0)) * nx + (ng + (ig - B.gi0)));
        };
        auto it = T.by_rank.find(me);
        if (it != T.by_rank.end())
            for (int kb : it->second) {
                const HaloBlock &B = T.blk[kb];
                if (B.gj0 + B.gny - 1 != NY) continue;            // not a top-row block
                {   // the north-west corner ghost cell
                    int ig = B.gi0 - 1;
                    if (ig < 1) ig += NX;
                    if (ig != NX / 2 && ig != NX) {
                        int im = NX - ig + 2;
                        if (im > NX) im -= NX;
                        int owner = -1;
                        const int32_t src = cell_below(im, owner);
                        if (owner != me) plan.stress_remote = true;
                        plan.stress_corner_dst.push_back((int32_t)((size_t)B.local * plane + (size_t)(ng + B.gny) * nx + (ng - 1)));
                        plan.stress_corner_src.push_back(owner == me ? src : -1);
                    }
                }
                const int j = ng + B.gny;                         // local row of global NY
                for (int i = 1; i <= B.gnx + 2 * ng; ++i) {
                    int ig = B.gi0 + (i - (ng + 1));
                    if (ig < 1) ig += NX;
                    if (ig > NX) ig -= NX;
                    int im = NX - ig + 2;
                    if (im > NX) im -= NX;
                    const int32_t dst = (int32_t)((size_t)B.local * plane + (size_t)(j - 1) * nx + (i - 1));
                    int owner = -1;
                    const int32_t src = cell_of(im, owner);
                    // (a partner on another rank, or in an eliminated land block -- where the shortcut of the call pairs does not hold:
                    // the symmetrisation then stays with the host)
                    if (owner != me) plan.stress_remote = true;
                    plan.stress_dst.push_back(dst);
                    plan.stress_src.push_back(owner == me ? src : -1);
                    if (i <= ng || i > ng + B.gnx) {              // an east-west ghost cell: image of its own array's cell
                        int own = -1;
                        const int32_t s2 = cell_of(ig, own);
                        if (own != me) plan.stress_remote = true;
                        plan.stress_own_dst.push_back(dst);
                        plan.stress_own_src.push_back(own == me ? s2 : -1);
                    }
                }
            }
    }
    if (tripole) {       // where the shifted copies are built: this rank's interior cells of row NY-1 whose block also holds row NY
        auto it = T.by_rank.find(me);
        if (it != T.by_rank.end())
            for (int kb : it->second) {
                const HaloBlock &B = T.blk[kb];
                if (B.gj0 + B.gny - 1 != d.ny_global || B.gny < 2) continue;
                const int j = ng + B.gny - 1;                     // local row of global NY-1
                for (int i = ng + 1; i <= ng + B.gnx; ++i)
                    plan.fold_shift_cells.push_back((int32_t)((size_t)B.local * plane + (size_t)(j - 1) * nx + (i - 1)));
            }
    }
    // ghost cells whose source block was eliminated: ice_HaloUpdate_stress writes the fill value
    // (srcBlock == 0, ice_boundary.F90:7643-7645) -- the same cells the velocity plan zero-fills
    for (size_t k = 0; k < plan.local_dst.size(); ++k)
        if (plan.local_src[k] < 0 && tripole) {
            bool dup = false;
            for (int32_t dd : plan.stress_dst) dup |= dd == plan.local_dst[k];
            if (!dup) { plan.stress_dst.push_back(plan.local_dst[k]); plan.stress_src.push_back(-1); }
        }
    return true;
}


// tripoleT (T-fold; ice_boundary.F90:1563-1622 offsets and symmetrisation, :1686-1722 copy-out): rows NY and NY+1 of every
// block, ghost columns included, take column NX-ig+1-ioffset of the rows NY-joffset and NY-1-joffset, offsets (ioffset,
// joffset) = centre (-1, 0), NE corner (0, 1), E face (0, 0), N face (-1, 1); centre and E-face fields lie ON the fold: their
// top row is made symmetric first (pairs i <-> NX-i+2, i = 2..NX/2, resp. i <-> NX+1-i, i = 1..NX/2) -- an entry then holds
// the pair in the reference's order (a = the lower column) and flip says which half the destination is
static void build_fold_list_tfold(const cice_evp_hip_dims &d, int loc, FoldList &L)
{
    const int NX = d.nx_global, NY = d.ny_global, nx = d.nx_block, ng = d.nghost;
    const size_t plane = (size_t)nx * d.ny_block;
    std::vector<int> owner((size_t)NX * 3, -1);              // interior cell holding global (ig, NY-2 .. NY)
    for (int b = 0; b < d.nblocks; ++b)
        for (int j = d.jlo[b]; j <= d.jhi[b]; ++j) {
            const int jg = d.jglob0[b] + (j - d.jlo[b]);
            if (jg < NY - 2 || jg > NY) continue;
            for (int i = d.ilo[b]; i <= d.ihi[b]; ++i) {
                const int ig = d.iglob0[b] + (i - d.ilo[b]);
                owner[(size_t)(jg - (NY - 2)) * NX + (ig - 1)] = (int)((size_t)b * plane + (size_t)(j - 1) * nx + (i - 1));
            }
        }
    auto wrap = [&](int ig) {
        while (ig < 1) ig += NX;
        while (ig > NX) ig -= NX;
        return ig;
    };
    auto own = [&](int ig, int jg) { return owner[(size_t)(jg - (NY - 2)) * NX + (wrap(ig) - 1)]; };
    const int ioff = (loc == 0 || loc == 3) ? -1 : 0, joff = (loc == 1 || loc == 3) ? 1 : 0;
    const bool on_fold = (loc == 0 || loc == 2);
    for (int b = 0; b < d.nblocks; ++b)
        for (int j = d.jlo[b] - ng; j <= d.jhi[b] + ng; ++j) {
            const int jg = d.jglob0[b] + (j - d.jlo[b]);
            if (jg != NY && jg != NY + 1) continue;
            for (int i = d.ilo[b] - ng; i <= d.ihi[b] + ng; ++i) {
                const int ig = wrap(d.iglob0[b] + (i - d.ilo[b]));
                const int dd = (int)((size_t)b * plane + (size_t)(j - 1) * nx + (i - 1));
                const int m = wrap(NX - ig + 1 - ioff);
                const int jsrc = (jg == NY ? NY : NY - 1) - joff;
                int a = own(m, jsrc), bb = -1, fl = 1;
                if (on_fold && jg == NY && m != ig) {          // a pair of the symmetrised row
                    const int lo = std::min(ig, m), hi = std::max(ig, m);
                    a = own(lo, NY);
                    bb = own(hi, NY);
                    if (bb < 0) bb = -2;                         // partner's block eliminated: the buffer holds 0
                    fl = ig == lo ? 0 : 1;
                }
                L.dst.push_back(dd); L.a.push_back(a); L.b.push_back(bb); L.flip.push_back((uint8_t)fl);
            }
        }
}

void build_fold_list(const cice_evp_hip_dims &d, int loc, FoldList &L)
{
    L = FoldList();
    if (d.ns_boundary_type == CICE_EVP_BND_TRIPOLET) { build_fold_list_tfold(d, loc, L); return; }
    const int NX = d.nx_global, NY = d.ny_global, nx = d.nx_block, ng = d.nghost;
    const size_t plane = (size_t)nx * d.ny_block;
    std::vector<int> owner((size_t)NX * 2, -1);              // interior cell holding global (ig, NY-1) / (ig, NY)
    for (int b = 0; b < d.nblocks; ++b)
        for (int j = d.jlo[b]; j <= d.jhi[b]; ++j) {
            const int jg = d.jglob0[b] + (j - d.jlo[b]);
            if (jg < NY - 1 || jg > NY) continue;
            for (int i = d.ilo[b]; i <= d.ihi[b]; ++i) {
                const int ig = d.iglob0[b] + (i - d.ilo[b]);
                owner[(size_t)(jg - (NY - 1)) * NX + (ig - 1)] = (int)((size_t)b * plane + (size_t)(j - 1) * nx + (i - 1));
            }
        }
    auto wrap = [&](int ig) {
        while (ig < 1) ig += NX;
        while (ig > NX) ig -= NX;
        return ig;
    };
    auto own = [&](int ig, int row) { return owner[(size_t)row * NX + (wrap(ig) - 1)]; };   // row 0: NY-1, 1: NY
    auto add = [&](int dd, int aa, int bb, int fl) { L.dst.push_back(dd); L.a.push_back(aa); L.b.push_back(bb); L.flip.push_back((uint8_t)fl); };
    // a point ON the fold is averaged with its partner even when the partner's block was eliminated (the buffer holds 0)
    auto pair = [&](int dd, int ia, int ib, int fl) { const int pb = own(ib, 1); add(dd, own(ia, 1), pb >= 0 ? pb : -2, fl); };
    for (int b = 0; b < d.nblocks; ++b)
        for (int j = d.jlo[b] - ng; j <= d.jhi[b] + ng; ++j) {
            const int jg = d.jglob0[b] + (j - d.jlo[b]);
            if (jg != NY && jg != NY + 1) continue;
            for (int i = d.ilo[b] - ng; i <= d.ihi[b] + ng; ++i) {
                const int ig = wrap(d.iglob0[b] + (i - d.ilo[b]));
                const int dd = (int)((size_t)b * plane + (size_t)(j - 1) * nx + (i - 1));
                if (jg == NY) {
                    if (loc == 1) {                           // NE corner: pairs i <-> NX-i, poles NX/2 and NX
                        if (ig == NX / 2 || ig == NX) add(dd, own(ig, 1), -1, 1);
                        else if (ig < NX / 2) pair(dd, ig, NX - ig, 0);
                        else pair(dd, NX - ig, ig, 1);
                    } else if (loc == 3) {                    // N face: pairs i <-> NX+1-i
                        if (ig <= NX / 2) pair(dd, ig, NX + 1 - ig, 0);
                        else pair(dd, NX + 1 - ig, ig, 1);
                    }
                    continue;                                 // centre / E face: the top row is an ordinary row
                }
                // ghost row NY+1: mirror with offsets (0,0) centre, (1,1) NE corner, (1,0) E face, (0,1) N face
                const int is = (loc == 0 || loc == 3) ? NX - ig + 1 : NX - ig;
                const int row = (loc == 0 || loc == 2) ? 1 : 0;
                add(dd, own(is, row), -1, 1);
            }
        }
}

namespace {
// The cell a position's value comes from: start at the nearest interior cell of the window's block and walk, x first, then
// y, one array cell at a time.  Stepping onto a ghost cell that mirrors an interior cell continues FROM that interior cell
// (through periodic boundaries and into other blocks); a ghost cell nothing is copied into (closed boundary, eliminated
// neighbour) is an array cell like any other and the walk goes on through it while it stays inside the block's array --
// so a position outside the domain names the ghost cell that IS the array neighbour of the cells next to it (a position
// reached through a periodic wrap used to name the block's own corner ghost cell instead: the same "outside", but not the
// cell the reference reads there, and its static arrays need not agree -- round 5, the on-chip resident C-grid kernel).
struct WindowWalk {
    const cice_evp_hip_dims &d;
    int nxb, nyb;
    long plane;
    std::vector<int> owner;
    WindowWalk(const cice_evp_hip_dims &d_, const HaloPlan &P) : d(d_), nxb(d_.nx_block), nyb(d_.ny_block), plane((long)d_.nx_block * d_.ny_block)
    {
        owner.assign((size_t)plane * d.nblocks, -1);
        for (size_t k = 0; k < P.local_dst.size(); ++k) owner[P.local_dst[k]] = P.local_src[k];
    }
    bool interior(int b, int i, int j) const { return i >= d.ilo[b] && i <= d.ihi[b] && j >= d.jlo[b] && j <= d.jhi[b]; }
    long walk(int b, int i, int j, int ti, int tj) const               // from interior (b, i, j) by (ti, tj) steps
    {
        bool stat = false;
        auto step = [&](int di, int dj) {
            const int ni = i + di, nj = j + dj;
            if (ni < 1 || ni > nxb || nj < 1 || nj > nyb) return;      // (beyond the array: stay -- two steps outside a closed boundary)
            i = ni; j = nj;
            if (interior(b, i, j)) { stat = false; return; }
            const long c = (long)b * plane + (long)(j - 1) * nxb + (i - 1);
            if (owner[c] >= 0) {
                const long o = owner[c];
                b = (int)(o / plane);
                j = (int)((o % plane) / nxb) + 1;
                i = (int)((o % plane) % nxb) + 1;
                stat = false;
            } else {
                stat = true;
            }
        };
        for (; ti != 0; ti -= (ti > 0 ? 1 : -1)) step(ti > 0 ? 1 : -1, 0);
        for (; tj != 0; tj -= (tj > 0 ? 1 : -1)) step(0, tj > 0 ? 1 : -1);
        const long c = (long)b * plane + (long)(j - 1) * nxb + (i - 1);
        return stat ? -1 - c : c;
    }
    long at(int b, int i, int j) const                                 // window position (i, j) in block b's index space
    {
        const int ic = std::min(std::max(i, d.ilo[b]), d.ihi[b]);
        const int jc = std::min(std::max(j, d.jlo[b]), d.jhi[b]);
        return walk(b, ic, jc, i - ic, j - jc);
    }
};
}   // namespace

void build_window_table(const cice_evp_hip_dims &d, const HaloPlan &P, int OX, int OY, int strip, std::vector<int32_t> &tiles,
                        std::vector<int32_t> &tab, int extra)
{
    const int nxb = d.nx_block, nyb = d.ny_block;
    const long plane = (long)nxb * nyb;
    const WindowWalk W(d, P);
    tiles.clear();
    tab.clear();
    strip = std::max(1, strip);
    for (int b = 0; b < d.nblocks; ++b)
        for (long is0 = d.ilo[b]; is0 <= d.ihi[b]; is0 += (long)strip * (OX - 3))
            for (int j0 = d.jlo[b]; j0 <= d.jhi[b]; j0 += OY - 3)
                for (long i0 = is0; i0 <= d.ihi[b] && i0 < is0 + (long)strip * (OX - 3); i0 += OX - 3) {
                    bool regular = true;
                    // (extra = 1: one more row and column of positions per window, same owned range -- the on-chip resident
                    // kernel's velocity tile, evp_cgrid_res.hip)
                    for (int ty = 0; ty < OY + extra; ++ty)
                        for (int tx = 0; tx < OX + extra; ++tx) {
                            const int i = (int)i0 - 2 + tx, j = j0 - 2 + ty;
                            const long r = W.at(b, i, j);
                            tab.push_back((int32_t)r);
                            regular = regular && i >= 1 && i <= nxb && j >= 1 && j <= nyb &&
                                      r == (long)b * plane + (long)(j - 1) * nxb + (i - 1);
                        }
                    tiles.push_back(b);
                    tiles.push_back((int32_t)i0);
                    tiles.push_back(j0);
                    tiles.push_back(regular ? 1 : 0);
                }
}

void strip_zones(const cice_evp_hip_dims &d, const std::vector<int32_t> &tiles, int ex, int ey, const int *img_slot, std::vector<StripZone> &zones)
{
    const int nt = (int)(tiles.size() / 4), sx = ex - 3, sy = ey - 3;
    zones.clear();
    for (int b = 0; b < d.nblocks; ++b) {
        int i0 = 1 << 30, i1 = -1, j0 = 1 << 30, j1 = -1, cnt = 0;
        for (int w = 0; w < nt; ++w)
            if (tiles[4 * w] == b && tiles[4 * w + 3]) {
                i0 = std::min(i0, tiles[4 * w + 1]); i1 = std::max(i1, tiles[4 * w + 1]);
                j0 = std::min(j0, tiles[4 * w + 2]); j1 = std::max(j1, tiles[4 * w + 2]);
                ++cnt;
            }
        if (!cnt || (i1 - i0) % sx || (j1 - j0) % sy) continue;
        if (cnt != ((i1 - i0) / sx + 1) * ((j1 - j0) / sy + 1)) continue;       // (not a rectangle: cg_one keeps the block)
        if (i1 + sx - i0 < 62) continue;                                          // (narrower than a strip)
        // (cells with ghost images -- the block's outermost interior cells -- never lie inside: the marched kernel has no pushes)
        bool images = false;
        for (int j = j0; j <= j1 + sy - 1 && !images && img_slot; ++j)
            for (int i = i0; i <= i1 + sx - 1 && !images; ++i)
                images = img_slot[(size_t)b * d.nx_block * d.ny_block + (size_t)(j - 1) * d.nx_block + (i - 1)] >= 0;
        if (images) continue;
        zones.push_back(StripZone{b, i0, i1, j0, j1});
    }
}

int strip_items(const std::vector<StripZone> &zones, int ex, int ey, int lo0, long slots, int seg_min, int seg, std::vector<int32_t> &items)
{
    const int sx = ex - 3, sy = ey - 3, sown = 62 - lo0;
    items.clear();
    long nstrips = 0, maxrows = 0;
    for (const StripZone &z : zones) { nstrips += (z.i1 - z.i0 + sx + sown - 1) / sown; maxrows = std::max<long>(maxrows, z.j1 - z.j0 + sy); }
    if (seg <= 0) {
        const long nseg_fit = std::max<long>(1, slots / std::max<long>(1, nstrips));
        seg = (int)std::max<long>(seg_min, (maxrows + nseg_fit - 1) / nseg_fit);
    }
    for (const StripZone &z : zones) {
        const int rows = z.j1 - z.j0 + sy, nseg = (rows + seg - 1) / seg;
        const int ilast = z.i1 + sx - 1;                       // last owned column of the rectangle
        for (int k = 0; k < nseg; ++k) {
            // (equal segments: rows / nseg, the remainder one row each to the first ones)
            const int ja = z.j0 + (int)((long)rows * k / nseg), jb = z.j0 + (int)((long)rows * (k + 1) / nseg) - 1;
            for (int i0 = z.i0; i0 <= ilast; i0 += sown) {
                // column of lane 2: the strip's first owned column on lane lo0, or further west if lane 61 would pass the rectangle
                const int c = std::min(i0 - (lo0 - 2), std::max(z.i0 - (lo0 - 2), ilast - 59));
                const int lo = 2 + (i0 - c), hi = std::min(61, 2 + (ilast - c));
                items.push_back(z.b); items.push_back(c); items.push_back(ja); items.push_back(jb);
                items.push_back(lo); items.push_back(hi);
            }
        }
    }
    return seg;
}

void strip_windows(const std::vector<StripZone> &zones, const std::vector<int32_t> &tiles, std::vector<uint8_t> &in_zone)
{
    const int nt = (int)(tiles.size() / 4);
    in_zone.assign((size_t)nt, 0);
    for (const StripZone &z : zones)
        for (int w = 0; w < nt; ++w)
            if (tiles[4 * w] == z.b && tiles[4 * w + 3] && tiles[4 * w + 1] >= z.i0 && tiles[4 * w + 1] <= z.i1 &&
                tiles[4 * w + 2] >= z.j0 && tiles[4 * w + 2] <= z.j1)
                in_zone[(size_t)w] = 1;
}

int cgres_dependencies(const cice_evp_hip_dims &d, bool tripole, const std::vector<int32_t> &tiles, const std::vector<int32_t> &tab,
                       std::vector<uint8_t> *pub, int *n_edges, int *n_oneway)
{
    constexpr int RX = 16, RY = 16, LW = RX + 1, NPOS = LW * (RY + 1);
    const int nt = (int)(tiles.size() / 4);
    const size_t ncell = (size_t)d.nblocks * d.nx_block * d.ny_block;
    auto jmax_of = [&](int w) { return tripole ? (int)(tiles[4 * w + 3] >> 16) : (int)d.jhi[tiles[4 * w]]; };
    auto foldwin = [&](int w) { return tripole && (tiles[4 * w + 3] & 1); };
    auto mine = [&](int w, int ex, int ey) {
        const int b = tiles[4 * w], i0 = tiles[4 * w + 1], j0 = tiles[4 * w + 2];
        return ex >= 2 && ex <= RX - 2 && ey >= 2 && ey <= RY - 2 && i0 - 2 + ex <= d.ihi[b] && j0 - 2 + ey <= jmax_of(w);
    };
    std::vector<int32_t> owner(ncell, -1);
    for (int w = 0; w < nt; ++w)
        for (int e = 0; e < NPOS; ++e)
            if (mine(w, e % LW, e / LW)) {
                const int sc = tab[(size_t)w * NPOS + e];
                if (sc >= 0 && (size_t)sc < ncell) owner[(size_t)sc] = w;
            }
    if (pub) pub->assign(ncell, 0);
    std::vector<std::pair<int, int>> edges;
    for (int w = 0; w < nt; ++w) {
        const int b = tiles[4 * w], i0 = tiles[4 * w + 1], j0 = tiles[4 * w + 2];
        const int last_ex = std::min(RX - 2, 2 + d.ihi[b] - i0), last_ey = std::min(RY - 2, 2 + jmax_of(w) - j0);
        for (int e = 0; e < NPOS - 1; ++e) {              // ((RX, RY), the one entry no level reads, is left out)
            const int ex = e % LW, ey = e / LW;
            const int sc = tab[(size_t)w * NPOS + e];
            if (mine(w, ex, ey) || sc < 0 || !cgres_in_reach(ex, ey, last_ex, last_ey, foldwin(w))) continue;
            if (pub) (*pub)[(size_t)sc] = 1;
            const int p = owner[(size_t)sc];
            if (p >= 0 && p != w) edges.emplace_back(w, p);
        }
    }
    std::sort(edges.begin(), edges.end());
    edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
    if (n_edges) *n_edges = (int)edges.size();
    // reads[w]: the windows w reads (edges is sorted by reader)
    std::vector<int> first((size_t)nt + 1, 0);
    for (const auto &e : edges) ++first[(size_t)e.first + 1];
    for (int w = 0; w < nt; ++w) first[(size_t)w + 1] += first[(size_t)w];
    int oneway = 0, unsafe = 0;
    std::vector<int> seen((size_t)nt, -1), frontier, next;
    int stamp = 0;
    for (const auto &e : edges) {
        if (std::binary_search(edges.begin(), edges.end(), std::make_pair(e.second, e.first))) continue;
        ++oneway;
        // w = e.first reads p = e.second and p does not read w: is there a chain p reads ... reads w of at most CGRES_SLOTS - 1?
        const int w = e.first, p = e.second;
        ++stamp;
        frontier.assign(1, p);
        seen[(size_t)p] = stamp;
        bool found = false;
        for (int len = 1; len <= CGRES_SLOTS - 1 && !found && !frontier.empty(); ++len) {
            next.clear();
            for (int x : frontier)
                for (int k = first[(size_t)x]; k < first[(size_t)x + 1] && !found; ++k) {
                    const int y = edges[(size_t)k].second;
                    if (y == w) found = true;
                    else if (seen[(size_t)y] != stamp) { seen[(size_t)y] = stamp; next.push_back(y); }
                }
            frontier.swap(next);
        }
        if (!found) ++unsafe;
    }
    if (n_oneway) *n_oneway = oneway;
    return unsafe;
}

// Windows of the on-chip resident C-grid kernel on a tripole (u-fold) grid (evp_cgrid_res.hip, template variant FOLD).  17 x 17
// positions per window, 13 x 13 owned as in build_window_table(..., extra = 1), except:
//  * the top window row of a block that touches the fold owns the block's last (up to) 11 rows, so that the fold row NY sits
//    at tile row tf <= 12 and three more tile rows remain; the window rows below it stop where it starts;
//  * tile rows tf+1 .. tf+3 of those windows hold a MIRRORED mini-tile in SOURCE orientation: global rows NY-2, NY-1, NY, tile
//    column tx <-> global column G0' + tx with G0' = NX - G0 - 15, G0 + tx = the global column of normal tile column tx.  A
//    normal fold-row position tx then faces the E-face / corner-type source at mirrored column 15 - tx and the centre / N-face
//    type source at 16 - tx (ice_boundary.F90:1626-1722: NX - ig for E faces and NE corners, NX - ig + 1 for centres and N
//    faces);
//  * tile rows above the mini-tile are unused (marked static, naming the window's first cell).
// tiles: (block, i0, j0, flags) with flags bit 0 = fold window, bits 8-15 = tf, bits 16-31 = last owned row (block index
// space); tiles2: (G0, NX, 0, 0).  Returns false (and says why) when a mirrored cell is not an interior cell of a block on
// this rank -- the resident kernel then is not used.
bool build_fold_window_table(const cice_evp_hip_dims &d, const HaloPlan &P, std::vector<int32_t> &tiles, std::vector<int32_t> &tiles2,
                             std::vector<int32_t> &tab, std::string &why)
{
    const int X = 16, OWN = 13, FOLDOWN = 11;
    const int nxb = d.nx_block;
    const long plane = (long)nxb * d.ny_block;
    const int NX = d.nx_global, NY = d.ny_global;
    const WindowWalk W(d, P);
    tiles.clear(); tiles2.clear(); tab.clear();
    std::vector<int32_t> cell((size_t)NX * NY, -1);                    // global (ig, jg) -> local interior cell
    for (int b = 0; b < d.nblocks; ++b)
        for (int j = d.jlo[b]; j <= d.jhi[b]; ++j)
            for (int i = d.ilo[b]; i <= d.ihi[b]; ++i) {
                const int ig = d.iglob0[b] + (i - d.ilo[b]), jg = d.jglob0[b] + (j - d.jlo[b]);
                if (ig >= 1 && ig <= NX && jg >= 1 && jg <= NY) cell[(size_t)(jg - 1) * NX + (ig - 1)] = (int32_t)(b * plane + (long)(j - 1) * nxb + (i - 1));
            }
    auto wrap = [&](long ig) { ig = (ig - 1) % NX; if (ig < 0) ig += NX; return (int)ig + 1; };
    for (int b = 0; b < d.nblocks; ++b) {
        const bool top = d.jglob0[b] + (d.jhi[b] - d.jlo[b]) == NY;
        const int jtop = top ? std::max(d.jlo[b], d.jhi[b] - (FOLDOWN - 1)) : d.jhi[b] + 1;   // first row of the fold windows
        if (top && d.jhi[b] - d.jlo[b] + 1 < 3) { why = "a block at the fold has fewer than three rows"; return false; }
        for (int j0 = d.jlo[b]; j0 <= d.jhi[b]; j0 = (j0 < jtop && j0 + OWN >= jtop) ? jtop : j0 + OWN) {
            const bool fw = top && j0 == jtop;
            const int jmax = fw ? d.jhi[b] : std::min(j0 + OWN - 1, jtop - 1);
            const int tf = fw ? 2 + (d.jhi[b] - j0) : 0;
            for (int i0 = d.ilo[b]; i0 <= d.ihi[b]; i0 += OWN) {
                const long G0 = (long)d.iglob0[b] + (i0 - 2 - d.ilo[b]);
                const long G0m = (long)NX - G0 - 15;
                const int32_t dead = (int32_t)(-1 - (b * plane + (long)(j0 - 1) * nxb + (i0 - 1)));
                for (int ty = 0; ty <= X; ++ty)
                    for (int tx = 0; tx <= X; ++tx) {
                        if (!fw || ty <= tf) { tab.push_back((int32_t)W.at(b, i0 - 2 + tx, j0 - 2 + ty)); continue; }
                        if (ty > tf + 3) { tab.push_back(dead); continue; }
                        const int jg = NY - (tf + 3 - ty), ig = wrap(G0m + tx);
                        const int32_t c = jg >= 1 ? cell[(size_t)(jg - 1) * NX + (ig - 1)] : -1;
                        if (c < 0) { why = "a cell mirrored across the fold is not on this rank"; return false; }
                        tab.push_back(c);
                    }
                tiles.push_back(b); tiles.push_back(i0); tiles.push_back(j0);
                tiles.push_back((fw ? 1 : 0) | (tf << 8) | (jmax << 16));
                tiles2.push_back((int32_t)G0); tiles2.push_back(NX); tiles2.push_back(0); tiles2.push_back(0);
            }
            if (fw) break;
        }
    }
    return true;
}
