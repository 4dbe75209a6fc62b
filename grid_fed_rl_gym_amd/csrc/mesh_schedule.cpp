// mesh_schedule.cpp -- see mesh_schedule.h.  Pure host C++.
#include "mesh_schedule.h"

#include <algorithm>
#include <map>
#include <set>
#include <utility>

namespace {



struct Target {                          // what a message adds to: the diagonal block + right-hand side of a bus, or an off-diagonal block
  int cons = -1;                         // the pivot that pulls it
  int units = 0;                         // 16-byte-per-lane units: CQ 3, M 2
  std::map<int, int> per_level;          // producers per level so far
  std::vector<int> acc;                  // accumulator ids, by rank inside a level
};

struct Acc { int units = 0, first = 1 << 30, cons_level = 0, off = -1; };

}  // namespace

void gs_mesh_schedule(const HostTopology& ht, int NW, int NI, int IW, int region_base, int slot_bytes, int acc_cap, int unit_budget, MeshSchedule& S) {
  S = MeshSchedule();
  S.NW = NW; S.NI = NI; S.IW = IW; S.HV = 64 / IW;
  const int n = ht.n, HV = S.HV;
  S.unit_bytes = 16 * IW; S.zero_off = 0; S.dummy_off = 3 * S.unit_bytes; S.body_off = 6 * S.unit_bytes;
  auto fail = [&](const std::string& w) { S.ok = false; S.why = w; };
  if (HV != 8) return fail("the meshed member is built for 8 instances per workgroup");
  if (acc_cap < 1) acc_cap = 1;
  if (acc_cap > GS_MESH_ACC) acc_cap = GS_MESH_ACC;
  std::vector<char> active(n);
  for (int i = 0; i < n; ++i) active[i] = ht.th_free[i] || ht.vm_free[i];

  // ---- minimum-degree elimination on the active buses (ties: lowest index), as topology.cpp orders the first-generation LU ----
  std::vector<std::set<int>> g(n);
  for (int i = 0; i < n; ++i)
    for (int p = ht.row_ptr[i]; p < ht.row_ptr[i + 1]; ++p) {
      const int j = ht.col[p];
      if (j != i && active[i] && active[j]) g[i].insert(j);
    }
  std::vector<int> order, pos_of(n, -1);
  std::vector<std::vector<int>> nbrs(n);
  {
    std::vector<char> gone(n, 0);
    for (int step = 0; step < ht.n_active; ++step) {
      int k = -1; size_t best = (size_t)-1;
      for (int i = 0; i < n; ++i) if (active[i] && !gone[i] && g[i].size() < best) { best = g[i].size(); k = i; }
      if (k < 0) break;
      pos_of[k] = (int)order.size(); order.push_back(k);
      nbrs[k].assign(g[k].begin(), g[k].end());
      for (int i : nbrs[k]) for (int j : nbrs[k]) if (i != j) g[i].insert(j);
      for (int j : nbrs[k]) g[j].erase(k);
      gone[k] = 1;
      S.max_degree = std::max(S.max_degree, (int)nbrs[k].size());
    }
  }
  S.n_pivots = (int)order.size();
  if (S.max_degree > HV) return fail("a pivot of the block LU has more than " + std::to_string(HV) + " neighbours when it is eliminated");

  // ---- levels.  First as early as the dependencies allow, a pivot delayed only while one of its targets already has acc_cap
  // producers in the level.  Then pivots are moved LATER inside their windows while that helps: a message occupies its accumulator
  // from its producer's level to its consumer's, so a leaf eliminated eight levels before the bus it reports to holds 48 bytes
  // per instance for nothing; what is minimised is, in this order, the excess of the peak over `unit_budget`, the excess of the
  // row count over `row_budget`, the levels' critical path (rows per level / waves), the peak, and the sum of squares of the
  // per-level occupancy.  (The benchmark's 123-bus feeder with 26 loops: 392 -> 289 units, i.e. two workgroups per CU.)
  auto target_cons = [&](int i, int j) { return (i == j) ? i : (pos_of[i] < pos_of[j] ? i : j); };
  std::vector<int> level(n, 0);
  {
    std::vector<int> ready(n, 0);
    std::map<std::pair<int, int>, std::map<int, int>> cnt;
    for (int k : order) {
      int lv = ready[k];
      for (bool again = true; again;) {
        again = false;
        for (int i : nbrs[k]) for (int j : nbrs[k]) {
          auto& pl = cnt[{i, j}];
          auto it = pl.find(lv);
          if (it != pl.end() && it->second >= acc_cap) { ++lv; again = true; }
        }
      }
      level[k] = lv;
      for (int i : nbrs[k]) { ready[i] = std::max(ready[i], lv + 1); for (int j : nbrs[k]) ++cnt[{i, j}][lv]; }
    }
  }
  struct Score { long long over_units, over_rows, critical, peak, sumsq; int maxA; };
  auto better = [](const Score& a, const Score& b) {
    if (a.over_units != b.over_units) return a.over_units < b.over_units;
    if (a.over_rows != b.over_rows) return a.over_rows < b.over_rows;
    if (a.critical != b.critical) return a.critical < b.critical;
    if (a.peak != b.peak) return a.peak < b.peak;
    return a.sumsq < b.sumsq;
  };
  auto pack_rows = [&](const std::vector<int>& lev, int NLv, std::vector<int>* rows_per_level) {
    int total = 0;
    for (int L = 0; L < NLv; ++L) {
      std::vector<int> sizes;
      for (int k : order) if (lev[k] == L) sizes.push_back(std::max(1, (int)nbrs[k].size()));
      if (L == 0) sizes.push_back(1);                                       // the slack's position
      std::sort(sizes.begin(), sizes.end(), std::greater<int>());
      std::vector<int> bins;
      for (int g : sizes) {
        bool put = false;
        for (int& b : bins) if (b + g <= HV) { b += g; put = true; break; }
        if (!put) bins.push_back(g);
      }
      if (rows_per_level) rows_per_level->push_back((int)bins.size());
      total += (int)bins.size();
    }
    return total;
  };
  auto evaluate = [&](const std::vector<int>& lev) {
    int NLv = 0;
    for (int k : order) NLv = std::max(NLv, lev[k] + 1);
    std::map<std::pair<int, int>, std::map<int, int>> cnt;
    for (int k : order) for (int i : nbrs[k]) for (int j : nbrs[k]) ++cnt[{i, j}][lev[k]];
    std::vector<long long> live(NLv + 1, 0);
    Score sc{0, 0, 0, 0, 0, 0};
    for (auto& kv : cnt) {
      const int i = kv.first.first, j = kv.first.second, units = (i == j) ? 3 : 2, cl = lev[target_cons(i, j)];
      int A = 0;
      for (auto& pl : kv.second) A = std::max(A, pl.second);
      sc.maxA = std::max(sc.maxA, A);
      for (int a = 0; a < A; ++a) {
        int first = 1 << 30;
        for (auto& pl : kv.second) if (pl.second > a) first = std::min(first, pl.first);
        for (int L = first; L <= cl && L <= NLv; ++L) live[L] += units;
      }
    }
    for (long long x : live) { sc.peak = std::max(sc.peak, x); sc.sumsq += x * x; }
    std::vector<int> rpl;
    const int rows = pack_rows(lev, NLv, &rpl);
    for (int r : rpl) sc.critical += (r + NW - 1) / NW;
    sc.over_units = std::max<long long>(0, sc.peak - unit_budget);
    sc.over_rows = std::max(0, rows - NW * NI);
    return sc;
  };
  if (unit_budget > 0) {
    Score best = evaluate(level);
    for (int round = 0; round < 6; ++round) {
      bool improved = false;
      for (int k : order) {
        int hi = level[k];
        if (!nbrs[k].empty()) { hi = 1 << 30; for (int i : nbrs[k]) hi = std::min(hi, level[i] - 1); }
        const int keep = level[k];
        int pick = keep;
        for (int cand = keep + 1; cand <= hi; ++cand) {
          level[k] = cand;
          const Score sc = evaluate(level);
          if (sc.maxA <= acc_cap && better(sc, best)) { best = sc; pick = cand; improved = true; }
        }
        level[k] = pick;
      }
      if (!improved) break;
    }
  }
  for (int k : order) S.n_levels = std::max(S.n_levels, level[k] + 1);
  if (S.n_levels < 1) S.n_levels = 1;

  // ---- targets and accumulators: the producers of a target in one level get accumulators of their own (by rank); an accumulator's
  // first producer writes, the later ones (later levels) add ----
  std::map<std::pair<int, int>, Target> targets;          // (i, i): CQ of bus i; (i, j): block (i, j)
  auto target_of = [&](int i, int j) -> Target& {
    Target& t = targets[{i, j}];
    if (t.cons < 0) { t.cons = target_cons(i, j); t.units = (i == j) ? 3 : 2; }
    return t;
  };
  std::vector<Acc> accs;
  struct Msg { int prod, i, j, acc, rmw; };
  std::vector<Msg> msgs;
  for (int k : order)
    for (int i : nbrs[k]) for (int j : nbrs[k]) {
      Target& t = target_of(i, j);
      const int rank = t.per_level[level[k]]++;
      if (rank >= acc_cap) return fail("internal: more producers of one target in a level than accumulators");
      if (rank >= (int)t.acc.size()) { t.acc.push_back((int)accs.size()); Acc a; a.units = t.units; accs.push_back(a); }
      msgs.push_back({k, i, j, t.acc[rank], 0});
    }
  for (auto& m : msgs) accs[m.acc].first = std::min(accs[m.acc].first, level[m.prod]);
  for (auto& m : msgs) m.rmw = level[m.prod] > accs[m.acc].first ? 1 : 0;
  for (auto& kv : targets) for (int a : kv.second.acc) accs[a].cons_level = level[kv.second.cons];
  for (auto& m : msgs)
    if (level[m.prod] >= accs[m.acc].cons_level) return fail("internal: a message is produced at or after its consumer's level");
  S.n_messages = (int)msgs.size(); S.n_accumulators = (int)accs.size();

  // ---- LDS units of the accumulators: live from the first producer's level through the consumer's; reusable by producers of later levels ----
  {
    std::vector<int> by_first(accs.size());
    for (size_t a = 0; a < accs.size(); ++a) by_first[a] = (int)a;
    std::stable_sort(by_first.begin(), by_first.end(), [&](int x, int y) { return accs[x].first < accs[y].first; });
    std::vector<int> free_from;                               // per unit: the first level whose producers may write it again
    int top = 0;
    for (int a : by_first) {
      const int u = accs[a].units, f = accs[a].first;
      int at = -1;
      for (int o = 0; o + u <= top && at < 0; ++o) {          // lowest run of u units that are all free by level f
        bool fits = true;
        for (int q = 0; q < u; ++q) if (free_from[o + q] > f) { fits = false; break; }
        if (fits) at = o;
      }
      if (at < 0) {                                           // extend, taking along a free tail
        at = top;
        while (at > 0 && top - at + 1 <= u - 1 && free_from[at - 1] <= f) --at;
        top = at + u; free_from.resize(top, 0);
      }
      accs[a].off = at;
      for (int q = 0; q < u; ++q) free_from[at + q] = accs[a].cons_level + 1;
    }
    S.msg_units = std::max(top, n);                          // the x slots of the back substitution share the body: one unit per bus
  }
  S.region_bytes = S.body_off + S.msg_units * S.unit_bytes;
  const int ZERO = region_base + S.zero_off, DUMMY = region_base + S.dummy_off, BODY = region_base + S.body_off;
  auto acc_addr = [&](int a) { return BODY + accs[a].off * S.unit_bytes; };

  // ---- pull lists and outputs ----
  std::vector<std::vector<int>> cq_in(n);
  std::map<std::pair<int, int>, std::vector<int>> rw_in, cl_in;     // (pivot, neighbour) -> accumulator addresses of A(pivot, nbr) / A(nbr, pivot)
  for (auto& kv : targets) {
    const int i = kv.first.first, j = kv.first.second;
    for (int a : kv.second.acc) {
      if (i == j) cq_in[i].push_back(acc_addr(a));
      else if (kv.second.cons == i) rw_in[{i, j}].push_back(acc_addr(a));
      else cl_in[{j, i}].push_back(acc_addr(a));
    }
  }
  std::map<std::tuple<int, int, int>, std::pair<int, int>> out_of;   // (producer, i, j) -> (address, rmw)
  for (auto& m : msgs) out_of[{m.prod, m.i, m.j}] = {acc_addr(m.acc), m.rmw};

  // ---- groups -> rows (one level per row; first fit by decreasing size, a group = consecutive sub-groups) -> waves ----
  struct Group { int k, d, g, level, slackpos, gslot; };
  std::vector<Group> groups;
  for (int k : order) { const int d = (int)nbrs[k].size(); groups.push_back({k, d, std::max(d, 1), level[k], 0, 0}); }
  groups.push_back({ht.slack, 0, 1, 0, 1, 0});              // the slack bus: its share of the losses sum in the mismatch pass
  struct Row { int level; std::vector<std::pair<int, int>> lanes; int fill; int n_multi; };    // lanes: (group index, t) or (-1, 0)
  std::vector<Row> rows;
  std::vector<int> rows_of_level(S.n_levels, 0);
  for (int L = 0; L < S.n_levels; ++L) {
    std::vector<int> gl;
    for (size_t q = 0; q < groups.size(); ++q) if (groups[q].level == L) gl.push_back((int)q);
    std::stable_sort(gl.begin(), gl.end(), [&](int a, int b) { return groups[a].g > groups[b].g; });
    const size_t first_row = rows.size();
    for (int q : gl) {
      size_t r = first_row;
      while (r < rows.size() && rows[r].fill + groups[q].g > HV) ++r;
      if (r == rows.size()) { rows.push_back({L, std::vector<std::pair<int, int>>(HV, {-1, 0}), 0, 0}); ++rows_of_level[L]; }
      for (int t = 0; t < groups[q].g; ++t) rows[r].lanes[rows[r].fill + t] = {q, t};
      rows[r].fill += groups[q].g;
      if (groups[q].g > 1) groups[q].gslot = rows[r].n_multi++;             // (at most four groups of two or more in a row of eight)
    }
  }
  S.n_rows = (int)rows.size();
  std::vector<std::vector<int>> wave_rows(NW);
  {
    const int cap = (S.n_rows + NW - 1) / NW;
    int prev = 0;
    for (size_t r = 0; r < rows.size(); ++r) {
      int w = 0;
      for (int v = 1; v < NW; ++v) if (wave_rows[v].size() < wave_rows[w].size()) w = v;
      // consecutive levels of one row each (the sequential tail of the elimination) go to DIFFERENT waves in turn: everything a
      // row does before it needs its level's messages -- its item, the voltages, P / Q calculated, the original blocks -- then
      // runs while another wave is still on the level before
      if (rows_of_level[rows[r].level] == 1) {
        int v = (prev + 1) % NW;
        for (int q = 0; q < NW && (int)wave_rows[v].size() >= cap; ++q) v = (v + 1) % NW;
        if ((int)wave_rows[v].size() < cap) w = v;
      }
      wave_rows[w].push_back((int)r); prev = w;
    }
    for (auto& v : wave_rows) S.max_rows_per_wave = std::max(S.max_rows_per_wave, (int)v.size());
  }
  if (S.max_rows_per_wave > NI) return fail("the elimination needs " + std::to_string(S.max_rows_per_wave) + " rows per wavefront (limit " + std::to_string(NI) + ")");

  // ---- items ----
  const int V_ONE = (n + 1) * slot_bytes, V_ZERO = n * slot_bytes;
  S.rowinfo.assign((size_t)NW * NI * 4, 0);
  S.items.assign((size_t)NW * NI * HV, MeshItem{});
  for (int w = 0; w < NW; ++w)
    for (int j = 0; j < NI; ++j) {
      int32_t* ri = &S.rowinfo[((size_t)w * NI + j) * 4];
      ri[0] = -1; ri[1] = 1; ri[2] = 0; ri[3] = 0;
      const bool have = j < (int)wave_rows[w].size();
      const Row* row = have ? &rows[wave_rows[w][j]] : nullptr;
      int g_row = 1, ncq = 0, nrw = 0, ncl = 0, nadj = 0;
      if (have) {
        ri[0] = row->level;
        for (int hv = 0; hv < HV; ++hv) {
          const auto [q, t] = row->lanes[hv];
          if (q < 0) continue;
          const Group& G = groups[q];
          g_row = std::max(g_row, G.g);
          nadj = std::max(nadj, ht.row_ptr[G.k + 1] - ht.row_ptr[G.k]);
          if (!G.slackpos) ncq = std::max(ncq, (int)cq_in[G.k].size());
          if (t < G.d) {
            const int jn = nbrs[G.k][t];
            auto a = rw_in.find({G.k, jn}); if (a != rw_in.end()) nrw = std::max(nrw, (int)a->second.size());
            auto c = cl_in.find({G.k, jn}); if (c != cl_in.end()) ncl = std::max(ncl, (int)c->second.size());
          }
        }
        if (ncq > GS_MESH_ACC || nrw > GS_MESH_ACC || ncl > GS_MESH_ACC) return fail("internal: pull list too long");
        ri[1] = g_row | (ncq << 8) | (nrw << 16) | (ncl << 24); ri[2] = nadj;
      }
      for (int hv = 0; hv < HV; ++hv) {
        MeshItem& it = S.items[((size_t)w * NI + j) * HV + hv];
        it.pos = (int32_t)(((size_t)w * NI + j) * HV + hv);
        it.vk_off = V_ONE; it.vj_off = V_ONE; it.xk_off = DUMMY; it.xj_off = ZERO; it.cq_off = DUMMY;
        it.flags = (hv << MESH_F_HV0_SHIFT) | (1 << MESH_F_G_SHIFT);
        it.ykj_g = 0.0; it.ykj_b = 0.0; it.ykk_g = 0.0; it.ykk_b = -1.0;       // an idle lane's diagonal block comes out as the identity
        for (int q = 0; q < 8; ++q) it.mout[q] = DUMMY;
        it.bus = -1; it.nbr = -1;
        it.adj_ptr = (int32_t)S.adj_off.size();
        int q = -1, t = 0;
        if (have) { q = row->lanes[hv].first; t = row->lanes[hv].second; }
        const Group* G = q >= 0 ? &groups[q] : nullptr;
        // pull lists (padded with the ZERO message) and the pivot bus's Ybus row (padded with the ZERO voltage slot)
        std::vector<int> lcq, lrw, lcl;
        if (G) {
          const int hv0 = hv - t;
          it.flags = (hv0 << MESH_F_HV0_SHIFT) | (t << MESH_F_T_SHIFT) | (G->g << MESH_F_G_SHIFT);
          it.pad[0] = G->gslot;
          it.bus = G->k; it.vk_off = G->k * slot_bytes;
          // every lane of the group forms the pivot's D_k, r_k, D_k^-1 and s_k for itself (the same instructions on the same
          // operands as lane 0: nothing to exchange): all of them carry the diagonal entry, the pull list and the Ybus row
          if (t == 0) it.flags |= G->slackpos ? MESH_F_SLACKPOS : MESH_F_PIVOT;
          it.ykk_g = ht.Gd[G->k]; it.ykk_b = ht.Bd[G->k];
          if (!G->slackpos) { if (t == 0) it.xk_off = BODY + G->k * S.unit_bytes; lcq = cq_in[G->k]; }
          if (t < G->d) {
            const int jn = nbrs[G->k][t];
            it.flags |= MESH_F_NBR; it.nbr = jn;
            it.vj_off = jn * slot_bytes; it.xj_off = BODY + jn * S.unit_bytes;
            for (int p = ht.row_ptr[G->k]; p < ht.row_ptr[G->k + 1]; ++p) if (ht.col[p] == jn) { it.ykj_g = ht.G[p]; it.ykj_b = ht.B[p]; }
            auto a = rw_in.find({G->k, jn}); if (a != rw_in.end()) lrw = a->second;
            auto c = cl_in.find({G->k, jn}); if (c != cl_in.end()) lcl = c->second;
            int rmw = 0;
            for (int t2 = 0; t2 < G->d; ++t2) {
              const auto o = out_of.at({G->k, jn, nbrs[G->k][t2]});
              it.mout[t2] = o.first; rmw |= o.second << t2;
              if (t2 == t) it.cq_off = o.first;
            }
            it.flags |= rmw << MESH_F_RMW_SHIFT;
          }
        }
        lcq.resize(GS_MESH_ACC, ZERO); lrw.resize(GS_MESH_ACC, ZERO); lcl.resize(GS_MESH_ACC, ZERO);
        for (int u = 0; u < GS_MESH_ACC; ++u) { it.cq_in[u] = lcq[u]; it.rw_in[u] = lrw[u]; it.cl_in[u] = lcl[u]; }
        int have_adj = 0;
        if (G)
          for (int p = ht.row_ptr[G->k]; p < ht.row_ptr[G->k + 1]; ++p, ++have_adj) {
            S.adj_off.push_back(ht.col[p] * slot_bytes); S.adj_y.push_back(ht.G[p]); S.adj_y.push_back(ht.B[p]);
          }
        for (; have_adj < nadj; ++have_adj) { S.adj_off.push_back(V_ZERO); S.adj_y.push_back(0.0); S.adj_y.push_back(0.0); }
      }
    }
  // ---- the packed form the kernel reads ----
  {
    std::map<std::pair<int, int>, int> pair_of;
    for (int i = 0; i < n; ++i)
      for (int p = ht.row_ptr[i]; p < ht.row_ptr[i + 1]; ++p) {
        const int j = ht.col[p];
        if (j <= i) continue;
        pair_of[{i, j}] = S.n_pairs++;
        S.ytab.push_back(ht.G[p]); S.ytab.push_back(ht.B[p]);
      }
    S.ytab.push_back(0.0); S.ytab.push_back(0.0);                                   // entry n_pairs: no branch
    for (int i = 0; i < n; ++i) { S.ytab.push_back(ht.Gd[i]); S.ytab.push_back(ht.Bd[i]); }
    for (int q = 0; q < 3; ++q) { S.ytab.push_back(0.0); S.ytab.push_back(-1.0); }   // ZERO, ONE, DUMMY slots
    std::vector<int> adj_ptr(n + 1, 0);
    for (int i = 0; i < n; ++i) {
      adj_ptr[i] = (int)S.adj_ent.size();
      for (int p = ht.row_ptr[i]; p < ht.row_ptr[i + 1]; ++p) {
        const int j = ht.col[p];
        if (j == i) continue;
        S.adj_ent.push_back(pair_of.at({std::min(i, j), std::max(i, j)}) | (j << 16));
      }
    }
    adj_ptr[n] = (int)S.adj_ent.size();
    S.adj_ent.push_back(S.n_pairs | (n << 16));            // the last entry: no branch, the ZERO voltage slot (what a lane reads beyond its bus's neighbours)
    if (S.n_pairs >= 65535 || S.adj_ent.size() >= 65535 || n + 3 >= 65535 || 6 + S.msg_units >= 65535) return fail("internal: a packed field overflows 16 bits");
    const int U = S.unit_bytes;
    auto unit_of = [&](int addr) { return (addr - region_base) / U; };
    S.packed.assign(S.items.size() * GS_MESH_WORDS, 0);
    S.rowinfo_packed = S.rowinfo;
    for (size_t q = 0; q < S.items.size(); ++q) {
      const MeshItem& it = S.items[q];
      int32_t* w = &S.packed[q * GS_MESH_WORDS];
      const int bus = it.vk_off / slot_bytes, nbr = it.vj_off / slot_bytes;
      w[GS_MESH_W_BUS_NBR] = bus | (nbr << 16);
      w[GS_MESH_W_FLAGS] = it.flags;
      int pair = S.n_pairs;
      if ((it.flags & MESH_F_NBR) && (it.ykj_g != 0.0 || it.ykj_b != 0.0)) pair = pair_of.at({std::min(bus, nbr), std::max(bus, nbr)});
      w[GS_MESH_W_PAIR_CQ] = pair | (unit_of(it.cq_off) << 16);
      for (int t = 0; t < 8; t += 2) w[GS_MESH_W_MOUT + t / 2] = unit_of(it.mout[t]) | (unit_of(it.mout[t + 1]) << 16);
      for (int u = 0; u < GS_MESH_ACC; u += 2) {
        w[GS_MESH_W_CQIN + u / 2] = unit_of(it.cq_in[u]) | (unit_of(it.cq_in[u + 1]) << 16);
        w[GS_MESH_W_CQIN + 2 + u / 2] = unit_of(it.rw_in[u]) | (unit_of(it.rw_in[u + 1]) << 16);
        w[GS_MESH_W_CQIN + 4 + u / 2] = unit_of(it.cl_in[u]) | (unit_of(it.cl_in[u + 1]) << 16);
      }
      w[GS_MESH_W_GSLOT] = it.pad[0];
      const bool lane0 = it.bus >= 0;                          // (every lane of a group carries its pivot's diagonal entry and Ybus row)
      const int dslot = lane0 ? bus : n + 1;
      w[GS_MESH_W_DIAG_ADJ] = dslot | ((lane0 ? adj_ptr[bus] : 0) << 16);
      const int nadj_lane = lane0 ? adj_ptr[bus + 1] - adj_ptr[bus] : 0;
      if (nadj_lane > 255) return fail("a bus has more than 255 neighbours");
      w[GS_MESH_W_FLAGS] |= nadj_lane << GS_MESH_F_NADJ_SHIFT;
      if (lane0) {
        int32_t* ri = &S.rowinfo_packed[(q / HV) * 4];
        if (nadj_lane > ri[3]) ri[3] = nadj_lane;
      }
    }
    for (size_t r = 0; r < S.rowinfo_packed.size() / 4; ++r) { S.rowinfo_packed[4 * r + 2] = S.rowinfo_packed[4 * r + 3]; S.rowinfo_packed[4 * r + 3] = 0; }
  }
  S.ok = true;
}
