// Host-only geometry and exchange plan of the marching path: see march_plan.h.
#include "march_plan.h"

#include <algorithm>
#include <map>

namespace {

struct Blk { int gi0, gj0, gnx, gny, owner; };

constexpr int PW = MARCH_PLAN_PAD;        // width of the overlap / the ring

// column x of a rank's rectangle (may lie up to P cells beyond it) -> (strip, lane) that holds it
inline void column_home(int x, int own, int nstrips, int &s, int &l)
{
    s = std::min(std::max(x, 0) / own, nstrips - 1);
    l = x - s * own + PW;
}

// duplicates of the owner lanes of a rank with `nxr` columns in strips of `own`: -1, or (strip << 8) | lane.
// false: some column would need two duplicates (then a narrower strip is tried)
bool dup_table(int nxr, int own, bool wrapx, std::vector<int32_t> &dup)
{
    const int ns = (nxr + own - 1) / own;
    dup.assign((size_t)ns * 64, -1);
    for (int sb = 0; sb < ns; ++sb) {
        const int cnt = std::min(own, nxr - sb * own);       // columns strip sb owns: lanes P .. P+cnt-1
        for (int l = 0; l < 64; ++l) {
            if (l >= PW && l < PW + cnt) continue;             // an owner
            if (!(l < PW || l < 2 * PW + cnt)) continue;       // beyond the P overlap lanes: nothing reads it
            int x = sb * own - PW + l;
            if (wrapx) { if (x < 0) x += nxr; else if (x >= nxr) x -= nxr; }
            if (x < 0 || x >= nxr) continue;                 // a halo column beyond the rectangle: filled by the exchange
            const int so = x / own, lo = PW + x % own;        // its owner
            int32_t &slot = dup[(size_t)so * 64 + lo];
            if (slot != -1) return false;
            slot = (sb << 8) | l;
        }
    }
    return true;
}

}  // namespace

bool build_march_plan(const cice_evp_hip_dims &d, int own_max, bool wrap_inside, int ext, MarchPlan &P)
{
    P = MarchPlan();
    const int me = d.rank;
    const int NX = d.nx_global, NY = d.ny_global;
    if (d.nghost != 1) { P.error = "nghost != 1"; return false; }
    if (d.ns_boundary_type == CICE_EVP_BND_TRIPOLE || d.ns_boundary_type == CICE_EVP_BND_TRIPOLET ||
        d.ns_boundary_type == CICE_EVP_BND_CYCLIC) {
        P.error = "north-south boundary is not closed";
        return false;
    }
    const bool ew_cyclic = d.ew_boundary_type == CICE_EVP_BND_CYCLIC;
    std::vector<Blk> blk;
    if (d.gi0 != nullptr && d.nblocks_tot > 0) {
        for (int k = 0; k < d.nblocks_tot; ++k) blk.push_back({d.gi0[k] - 1, d.gj0[k] - 1, d.gnx[k], d.gny[k], d.gowner[k]});
    } else {
        if (d.nranks != 1) { P.error = "global block table required when nranks > 1"; return false; }
        for (int b = 0; b < d.nblocks; ++b)
            blk.push_back({d.iglob0[b] - 1, d.jglob0[b] - 1, d.ihi[b] - d.ilo[b] + 1, d.jhi[b] - d.jlo[b] + 1, me});
    }
    // every rank's rectangle
    const int nranks = std::max(1, (int)d.nranks);
    P.all.assign(nranks, MarchRect());
    std::vector<long> area(nranks, 0);
    std::vector<int> x0(nranks, 1 << 30), y0(nranks, 1 << 30), x1(nranks, -1), y1(nranks, -1);
    for (const Blk &b : blk) {
        if (b.owner < 0) { P.error = "eliminated land blocks: the ranks' sub-domains are not rectangles"; return false; }
        if (b.owner >= nranks) { P.error = "block owner out of range"; return false; }
        area[b.owner] += (long)b.gnx * b.gny;
        x0[b.owner] = std::min(x0[b.owner], b.gi0); y0[b.owner] = std::min(y0[b.owner], b.gj0);
        x1[b.owner] = std::max(x1[b.owner], b.gi0 + b.gnx - 1); y1[b.owner] = std::max(y1[b.owner], b.gj0 + b.gny - 1);
    }
    own_max = std::min(64 - 2 * PW, std::max(4, own_max));
    for (int r = 0; r < nranks; ++r) {
        if (area[r] == 0) continue;
        MarchRect &R = P.all[r];
        R.gx0 = x0[r]; R.gy0 = y0[r]; R.nxr = x1[r] - x0[r] + 1; R.nyr = y1[r] - y0[r] + 1;
        if ((long)R.nxr * R.nyr != area[r]) { P.error = "a rank's blocks do not tile a rectangle"; return false; }
        if (R.nxr < 4 || R.nyr < 1) { P.error = "a rank's rectangle is too small"; return false; }
        R.ok = true;
    }
    if (!P.all[me].ok) { P.error = "this rank holds no blocks"; return false; }
    if (ext < 0 || (ext & 1)) { P.error = "ext must be even and >= 0"; return false; }
    // the rectangle a rank holds = its own cells + ext on every side with a neighbour (the same rule for every rank)
    struct Held { MarchRect E; int w, e, s, n; bool wrap; };
    auto held = [&](int r) -> Held {
        const MarchRect &R = P.all[r];
        Held H{R, 0, 0, 0, 0, false};
        H.wrap = wrap_inside && ew_cyclic && R.nxr == NX;                   // wraps inside: no neighbour in x
        if (!H.wrap) {
            if (R.gx0 > 0 || ew_cyclic) H.w = ext;
            if (R.gx0 + R.nxr < NX || ew_cyclic) H.e = ext;
        }
        if (R.gy0 > 0) H.s = ext;
        if (R.gy0 + R.nyr < NY) H.n = ext;
        H.E.gx0 = R.gx0 - H.w; H.E.gy0 = R.gy0 - H.s;
        H.E.nxr = R.nxr + H.w + H.e; H.E.nyr = R.nyr + H.s + H.n;
        return H;
    };
    // A neighbour between a rank and a CLOSED boundary must be wide enough for the rim and its P-cell ring: otherwise the
    // held rectangle reaches past the global boundary, those positions stay zero here while the owner advances its cells
    // from the caller's boundary ghost values (rect_to_block) -- different operands for the same cell.  Such a layout is
    // refused (the one-subcycle kernels run it); every rank reaches the same verdict from the same table.
    for (int r = 0; r < nranks; ++r) {
        if (!P.all[r].ok) continue;
        const MarchRect &R = P.all[r];
        const int dw = R.gx0, de = NX - (R.gx0 + R.nxr), ds = R.gy0, dn = NY - (R.gy0 + R.nyr);
        const bool thin_x = !ew_cyclic && ((dw > 0 && dw < ext + PW) || (de > 0 && de < ext + PW));
        const bool thin_y = (ds > 0 && ds < ext + PW) || (dn > 0 && dn < ext + PW);
        if (thin_x || thin_y) {
            P.error = "a rank lies closer to a closed boundary than its redundant rim + ring is wide";
            return false;
        }
    }
    const Held HM = held(me);
    P.owned = P.all[me];
    P.me = HM.E;
    P.ext_w = HM.w; P.ext_e = HM.e; P.ext_s = HM.s; P.ext_n = HM.n;
    P.wrapx = HM.wrap;
    // strips: the widest `own` for which every column of this rank has at most one duplicate -- and the last strip
    // holds at least P columns: with fewer, the first columns BEYOND the rectangle (which the exchange fills)
    // would sit in two strips, as overlap lanes of the last but one and of the last, and a received cell has one home
    // (column_home) plus the duplicate of a column inside the rectangle only
    bool found = false;
    for (int own = own_max; own >= 4 && !found; --own) {
        const int ns = (P.me.nxr + own - 1) / own;
        if (ns > 1 && P.me.nxr - (ns - 1) * own < PW) continue;
        if (dup_table(P.me.nxr, own, P.wrapx, P.dup)) { P.me.own = own; found = true; }
    }
    if (!found) { P.error = "no strip width gives every column a single duplicate"; return false; }
    P.me.nstrips = (P.me.nxr + P.me.own - 1) / P.me.own;

    auto owner_of = [&](int gx, int gy) -> int {
        for (int r = 0; r < nranks; ++r) {
            const MarchRect &R = P.all[r];
            if (R.ok && gx >= R.gx0 && gx < R.gx0 + R.nxr && gy >= R.gy0 && gy < R.gy0 + R.nyr) return r;
        }
        return -1;
    };
    std::map<int, MarchPeer> peers;
    const MarchRect &M = P.me;
    for (int D = 0; D < nranks; ++D) {
        if (!P.all[D].ok) continue;
        const Held HD = held(D);
        const MarchRect &E = HD.E;                                          // what D holds; its own cells start at (HD.w, HD.s)
        for (int y = -PW; y < E.nyr + PW; ++y)
            for (int x = -PW; x < E.nxr + PW; ++x) {
                const bool own_cell = x >= HD.w && x < E.nxr - HD.e && y >= HD.s && y < E.nyr - HD.n;
                if (own_cell) continue;
                int gx = E.gx0 + x;
                const int gy = E.gy0 + y;
                if (gy < 0 || gy >= NY) continue;                          // closed north / south
                if (gx < 0 || gx >= NX) {
                    if (!ew_cyclic) continue;
                    if (HD.wrap) continue;                                 // D reads those columns through the wrap of its strips
                    gx = ((gx % NX) + NX) % NX;
                }
                const int S = owner_of(gx, gy);
                if (S < 0) continue;
                if (D == me) {
                    MarchPeer &p = peers[S];
                    p.rank = S;
                    int s, l;
                    column_home(x, M.own, M.nstrips, s, l);
                    const int row = y + MARCH_PLAN_PAD;
                    p.recv_pos1.push_back((int32_t)(((long)row * M.nstrips + s) * 64 + l));
                    int32_t p2 = -1;
                    if (x >= 0 && x < M.nxr) {
                        const int32_t dd = P.dup[(size_t)s * 64 + l];
                        if (dd >= 0) p2 = (int32_t)(((long)row * M.nstrips + (dd >> 8)) * 64 + (dd & 255));
                    }
                    p.recv_pos2.push_back(p2);
                    p.recv_col.push_back(x);
                    p.recv_row.push_back(row);
                }
                if (S == me) {
                    MarchPeer &p = peers[D];
                    p.rank = D;
                    const int xs = gx - P.owned.gx0 + HM.w, ys = gy - P.owned.gy0 + HM.s;   // the source cell in what I hold
                    const int s = xs / M.own, l = PW + xs % M.own;
                    p.send_pos.push_back((int32_t)(((long)(ys + MARCH_PLAN_PAD) * M.nstrips + s) * 64 + l));
                    p.send_col.push_back(xs);
                }
            }
    }
    for (auto &kv : peers) {
        P.n_send += (int)kv.second.send_pos.size();
        P.n_recv += (int)kv.second.recv_pos1.size();
        P.peers.push_back(kv.second);
    }
    return true;
}
