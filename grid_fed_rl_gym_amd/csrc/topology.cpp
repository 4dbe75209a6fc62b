// topology.cpp -- see topology.h.  Reference behaviour being matched is cited inline
// (paths relative to /root/reference/grid_fed_rl/).
#include "topology.h"
#include "gs_internal.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <queue>
#include <set>
#include <utility>

namespace {

struct Cx { double re, im; };

// 1/(r + jx) with the operation order of CPython's complex division, which is what the
// reference's `1.0 / z` executes (environments/power_flow.py:63): scale by the larger
// component so that Ybus matches the reference bit for bit.
Cx reciprocal(double r, double x) {
  Cx y;
  const double ar = std::fabs(r), ax = std::fabs(x);
  if (ar >= ax) {
    const double ratio = x / r;
    const double denom = r + x * ratio;
    y.re = (1.0 + 0.0 * ratio) / denom;
    y.im = (0.0 - 1.0 * ratio) / denom;
  } else {
    const double ratio = r / x;
    const double denom = r * ratio + x;
    y.re = (1.0 * ratio + 0.0) / denom;
    y.im = (0.0 * ratio - 1.0) / denom;
  }
  return y;
}

}  // namespace

std::string gs_compile_topology(const gs_topology& t, int zero_z_mode, bool want_lu, bool skip_lu, HostTopology& o) {
  const int n = t.n, m = t.m;
  if (n <= 0 || m < 0) return "n must be > 0 and m >= 0";
  if (!t.bus_type || !t.v_set) return "bus_type / v_set missing";
  if (m > 0 && (!t.from_bus || !t.to_bus || !t.r || !t.x || !t.rating)) return "line arrays missing";
  o = HostTopology();
  o.n = n; o.m = m;

  // ---- lines: series admittance (power_flow.py:62-63, 345-346) ------------------------
  o.lfrom.assign(t.from_bus, t.from_bus + m);
  o.lto.assign(t.to_bus, t.to_bus + m);
  o.lrating.assign(t.rating, t.rating + m);
  o.lrating_inv.assign(m, 0.0);
  for (int k = 0; k < m; ++k) if (o.lrating[k] > 0.0) o.lrating_inv[k] = 1.0 / o.lrating[k];
  o.lyr.resize(m); o.lyi.resize(m);
  for (int k = 0; k < m; ++k) {
    if (o.lfrom[k] < 0 || o.lfrom[k] >= n || o.lto[k] < 0 || o.lto[k] >= n) return "line endpoint out of range";
    double r = t.r[k], x = t.x[k];
    if (!(std::hypot(r, x) > 1e-12)) {
      if (zero_z_mode == GS_ZERO_Z_EPSILON) { r = 1e-4; x = 1e-4; }
      else { o.lyr[k] = 0.0; o.lyi[k] = 0.0; continue; }   // open circuit, as coded
    }
    const Cx y = reciprocal(r, x);
    o.lyr[k] = y.re; o.lyi[k] = y.im;
  }

  // ---- Ybus, accumulated in line order (power_flow.py:57-71) ----------------------------
  std::vector<std::map<int, Cx>> rows(n);
  for (int i = 0; i < n; ++i) rows[i][i] = Cx{0.0, 0.0};
  auto add = [&](int i, int j, double sr, double si) {
    auto it = rows[i].find(j);
    if (it == rows[i].end()) it = rows[i].emplace(j, Cx{0.0, 0.0}).first;
    it->second.re += sr; it->second.im += si;
  };
  for (int k = 0; k < m; ++k) {
    if (o.lyr[k] == 0.0 && o.lyi[k] == 0.0) continue;   // adding an exact zero changes nothing
    const int i = o.lfrom[k], j = o.lto[k];
    add(i, j, -o.lyr[k], -o.lyi[k]);
    add(j, i, -o.lyr[k], -o.lyi[k]);
    add(i, i, o.lyr[k], o.lyi[k]);
    add(j, j, o.lyr[k], o.lyi[k]);
  }
  o.row_ptr.assign(n + 1, 0);
  o.Gd.resize(n); o.Bd.resize(n);
  for (int i = 0; i < n; ++i) {
    o.row_ptr[i] = (int)o.col.size();
    for (auto& kv : rows[i]) {
      o.col.push_back(kv.first); o.G.push_back(kv.second.re); o.B.push_back(kv.second.im);
      if (kv.first == i) { o.Gd[i] = kv.second.re; o.Bd[i] = kv.second.im; }
    }
  }
  o.row_ptr[n] = (int)o.col.size();
  o.nnz = (int)o.col.size();
  // ELL(8) + CSR remainder view of the same rows, same entry order
  const int K = GS_ELL_K;
  o.ell_col.assign((size_t)n * K, 0); o.ell_G.assign((size_t)n * K, 0.0); o.ell_B.assign((size_t)n * K, 0.0);
  o.rem_ptr.assign(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    o.rem_ptr[i] = (int)o.rem_col.size();
    for (int k = 0; k < K; ++k) o.ell_col[(size_t)i * K + k] = i;
    for (int p = o.row_ptr[i], k = 0; p < o.row_ptr[i + 1]; ++p, ++k) {
      if (k < K) { o.ell_col[(size_t)i * K + k] = o.col[p]; o.ell_G[(size_t)i * K + k] = o.G[p]; o.ell_B[(size_t)i * K + k] = o.B[p]; }
      else { o.rem_col.push_back(o.col[p]); o.rem_G.push_back(o.G[p]); o.rem_B.push_back(o.B[p]); }
    }
  }
  o.rem_ptr[n] = (int)o.rem_col.size();
  auto pos_of = [&](int i, int j) -> int {
    for (int p = o.row_ptr[i]; p < o.row_ptr[i + 1]; ++p) if (o.col[p] == j) return p;
    return -1;
  };

  // ---- bus classification (power_flow.py:123-141) ---------------------------------------
  int slack = -1;
  for (int i = 0; i < n; ++i) if (t.bus_type[i] == GS_BUS_SLACK) slack = i;   // the last one wins
  if (slack < 0) slack = 0;      // defaulted slack stays in the pq list, as coded
  o.slack = slack;
  o.th_free.resize(n); o.vm_free.resize(n); o.fixed_v.resize(n);
  o.v_set.assign(t.v_set, t.v_set + n);
  for (int i = 0; i < n; ++i) {
    o.th_free[i] = (i != slack) ? 1 : 0;
    o.vm_free[i] = (t.bus_type[i] == GS_BUS_PQ) ? 1 : 0;
    o.fixed_v[i] = (t.bus_type[i] == GS_BUS_SLACK || t.bus_type[i] == GS_BUS_PV) ? 1 : 0;
  }

  // ---- elimination forest over the active buses ------------------------------------------
  std::vector<char> active(n);
  o.n_active = 0;
  for (int i = 0; i < n; ++i) { active[i] = (o.th_free[i] || o.vm_free[i]); o.n_active += active[i]; }
  std::vector<std::vector<int>> adj(n);
  for (int i = 0; i < n; ++i)
    for (int p = o.row_ptr[i]; p < o.row_ptr[i + 1]; ++p) {
      const int j = o.col[p];
      if (j != i && active[i] && active[j]) { adj[i].push_back(j); }
    }
  std::vector<char> touches_inactive(n, 0);
  for (int i = 0; i < n; ++i)
    for (int p = o.row_ptr[i]; p < o.row_ptr[i + 1]; ++p)
      if (o.col[p] != i && !active[o.col[p]]) touches_inactive[i] = 1;

  o.parent.assign(n, -1); o.parent_pos.assign(n, -1);
  std::vector<int> depth(n, -1), comp(n, -1);
  int n_comp = 0;
  bool forest = true;
  {
    // components first, so that each can be rooted at a bus next to the slack
    for (int s = 0; s < n; ++s) {
      if (!active[s] || comp[s] >= 0) continue;
      std::vector<int> members; std::queue<int> q; q.push(s); comp[s] = n_comp;
      while (!q.empty()) { int u = q.front(); q.pop(); members.push_back(u);
        for (int v : adj[u]) if (comp[v] < 0) { comp[v] = n_comp; q.push(v); } }
      int root = members[0];
      for (int u : members) if (touches_inactive[u]) { root = u; break; }
      int64_t e2 = 0;
      for (int u : members) e2 += (int64_t)adj[u].size();
      if (e2 / 2 != (int64_t)members.size() - 1) forest = false;
      // BFS from the root
      std::queue<int> q2; q2.push(root); depth[root] = 0;
      while (!q2.empty()) { int u = q2.front(); q2.pop();
        for (int v : adj[u]) if (depth[v] < 0) { depth[v] = depth[u] + 1; o.parent[v] = u; o.parent_pos[v] = pos_of(v, u); q2.push(v); } }
      ++n_comp;
    }
  }
  o.is_forest = forest;
  if (forest) {
    int maxd = -1;
    for (int i = 0; i < n; ++i) if (active[i]) maxd = std::max(maxd, depth[i]);
    o.n_levels = maxd + 1;
    o.lvl_ptr.assign(o.n_levels + 1, 0);
    for (int lv = 0; lv < o.n_levels; ++lv) {          // level 0 = deepest
      o.lvl_ptr[lv] = (int)o.lvl_bus.size();
      const int d = maxd - lv;
      for (int i = 0; i < n; ++i) if (active[i] && depth[i] == d) o.lvl_bus.push_back(i);
      o.max_level_width = std::max(o.max_level_width, (int)o.lvl_bus.size() - o.lvl_ptr[lv]);
    }
    o.lvl_ptr[o.n_levels] = (int)o.lvl_bus.size();
    o.lvl_pos.assign(n, -1);
    for (int lv = 0; lv < o.n_levels; ++lv)
      for (int t = o.lvl_ptr[lv]; t < o.lvl_ptr[lv + 1]; ++t) o.lvl_pos[o.lvl_bus[t]] = t - o.lvl_ptr[lv];
    o.child_ptr.assign(n + 1, 0);
    std::vector<std::vector<int>> ch(n);
    for (int i = 0; i < n; ++i) if (active[i] && o.parent[i] >= 0) ch[o.parent[i]].push_back(i);
    for (int i = 0; i < n; ++i) { o.child_ptr[i] = (int)o.child_idx.size(); for (int c : ch[i]) o.child_idx.push_back(c); }
    o.child_ptr[n] = (int)o.child_idx.size();
  } else {
    o.parent.assign(n, -1); o.parent_pos.assign(n, -1);
    o.lvl_ptr.assign(1, 0); o.child_ptr.assign(n + 1, 0); o.lvl_pos.assign(n, -1);
  }

  // ---- FBS eligibility: whole network is a tree under the slack, all other buses pq --------
  o.fbs_parent.assign(n, -1); o.fbs_parent_pos.assign(n, -1);
  {
    bool ok = forest && t.bus_type[slack] == GS_BUS_SLACK && o.n_active == n - 1;
    if (!ok) o.fbs_why = "network must be radial with exactly one typed slack bus";
    for (int i = 0; ok && i < n; ++i)
      if (i != slack && t.bus_type[i] != GS_BUS_PQ) { ok = false; o.fbs_why = "FBS handles pq buses only"; }
    if (ok) {
      // every forest root must hang off the slack, and the slack may touch each tree once
      int64_t slack_deg = 0;
      for (int p = o.row_ptr[slack]; p < o.row_ptr[slack + 1]; ++p) if (o.col[p] != slack) ++slack_deg;
      if (slack_deg != n_comp) { ok = false; o.fbs_why = "network is not radial (loop through the slack, or an island)"; }
      for (int i = 0; ok && i < n; ++i) {
        if (i == slack) continue;
        if (o.parent[i] >= 0) { o.fbs_parent[i] = o.parent[i]; o.fbs_parent_pos[i] = o.parent_pos[i]; }
        else {
          const int p = pos_of(i, slack);
          if (p < 0) { ok = false; o.fbs_why = "island without a path to the slack"; break; }
          o.fbs_parent[i] = slack; o.fbs_parent_pos[i] = p;
        }
      }
    }
    o.fbs_ok = ok;
  }

  // ---- sparse block-LU schedule (meshed networks): minimum-degree order, symbolic fill -----
  if ((!forest || want_lu) && !skip_lu) {
    o.has_lu = true;
    std::vector<std::set<int>> g(n);
    for (int i = 0; i < n; ++i) for (int v : adj[i]) g[i].insert(v);
    std::map<std::pair<int, int>, int> slot;
    for (int i = 0; i < n; ++i)
      for (int v : adj[i]) {
        const int s = (int)slot.size();
        slot[{i, v}] = s;
        o.lu_orig_slot.push_back(s); o.lu_orig_i.push_back(i); o.lu_orig_j.push_back(v);
        o.lu_orig_pos.push_back(pos_of(i, v));
      }
    o.lu_n_orig = (int)slot.size();
    std::vector<char> gone(n, 0);
    o.lu_nb_ptr.push_back(0); o.lu_pair_ptr.push_back(0);
    for (int step = 0; step < o.n_active; ++step) {
      int k = -1; size_t best = (size_t)-1;
      for (int i = 0; i < n; ++i) if (active[i] && !gone[i] && g[i].size() < best) { best = g[i].size(); k = i; }
      std::vector<int> nb(g[k].begin(), g[k].end());
      o.lu_piv_bus.push_back(k);
      for (int j : nb) {
        o.lu_nb_bus.push_back(j);
        o.lu_nb_kj.push_back(slot.at({k, j}));
        o.lu_nb_jk.push_back(slot.at({j, k}));
      }
      for (int i : nb)
        for (int j : nb) {
          o.lu_pair_ik.push_back(slot.at({i, k}));
          o.lu_pair_kj.push_back(slot.at({k, j}));
          if (i == j) { o.lu_pair_ij.push_back(-(1 + i)); }
          else {
            auto it = slot.find({i, j});
            if (it == slot.end()) { it = slot.emplace(std::make_pair(i, j), (int)slot.size()).first; g[i].insert(j); }
            o.lu_pair_ij.push_back(it->second);
          }
        }
      o.lu_nb_ptr.push_back((int)o.lu_nb_bus.size());
      o.lu_pair_ptr.push_back((int)o.lu_pair_ik.size());
      for (int j : nb) g[j].erase(k);
      gone[k] = 1;
    }
    o.lu_n_piv = o.n_active;
    o.lu_n_slots = (int)slot.size();
    o.lu_n_pairs = (int64_t)o.lu_pair_ik.size();
    {
      std::vector<int> dep(n, 0);
      o.lu_piv_level.assign(o.lu_n_piv, 0);
      for (int t = 0; t < o.lu_n_piv; ++t) {
        const int lv = dep[o.lu_piv_bus[t]];
        o.lu_piv_level[t] = lv;
        o.lu_n_levels = std::max(o.lu_n_levels, lv + 1);
        for (int q = o.lu_nb_ptr[t]; q < o.lu_nb_ptr[t + 1]; ++q) dep[o.lu_nb_bus[q]] = std::max(dep[o.lu_nb_bus[q]], lv + 1);
      }
    }
  }

  // ---- dense unknown numbering (power_flow.py:232-240, 291) -----------------------------------
  o.dn_th_idx.assign(n, -1); o.dn_vm_idx.assign(n, -1);
  {
    int k = 0;
    for (int i = 0; i < n; ++i) if (o.th_free[i]) o.dn_th_idx[i] = k++;
    for (int i = 0; i < n; ++i) if (o.vm_free[i]) o.dn_vm_idx[i] = k++;
    o.dn_N = k;
  }

  // ---- per-bus device lists (accumulation order of grid_env.py:689-718) --------------------
  auto bucket = [&](int count, const int32_t* bus, std::vector<int32_t>& ptr, std::vector<int32_t>& idx) -> bool {
    ptr.assign(n + 1, 0); idx.clear();
    std::vector<std::vector<int>> at(n);
    for (int k = 0; k < count; ++k) { if (bus[k] < 0 || bus[k] >= n) return false; at[bus[k]].push_back(k); }
    for (int i = 0; i < n; ++i) { ptr[i] = (int)idx.size(); for (int k : at[i]) idx.push_back(k); }
    ptr[n] = (int)idx.size();
    return true;
  };
  if (!bucket(t.n_loads, t.load_bus, o.bl_ptr, o.bl_idx)) return "load bus out of range";
  if (!bucket(t.n_gens, t.gen_bus, o.bg_ptr, o.bg_idx)) return "generator bus out of range";
  if (!bucket(t.n_bats, t.bat_bus, o.bb_ptr, o.bb_idx)) return "battery bus out of range";
  o.load_base.assign(t.load_base, t.load_base + t.n_loads);
  o.load_q.resize(t.n_loads);
  for (int l = 0; l < t.n_loads; ++l) o.load_q[l] = t.load_base[l] * std::tan(std::acos(t.load_pf[l]));   // base.py:283
  o.gen_kind.assign(t.gen_kind, t.gen_kind + t.n_gens);
  o.gen_cap.assign(t.gen_cap, t.gen_cap + t.n_gens);
  o.gen_p0.assign(t.gen_p0, t.gen_p0 + t.n_gens);
  o.gen_p1.assign(t.gen_p1, t.gen_p1 + t.n_gens);
  o.gen_p2.assign(t.gen_p2, t.gen_p2 + t.n_gens);
  o.bat_cap.assign(t.bat_cap, t.bat_cap + t.n_bats);
  o.bat_rating.assign(t.bat_rating, t.bat_rating + t.n_bats);
  o.bat_eff.assign(t.bat_eff, t.bat_eff + t.n_bats);
  return "";
}
