// graph.cpp — the stages behind findConsensusDRs: the spacer graph of every DR group and
// crass's output files, as plain host C++ over the flat hand-off (<= 10^4 reads per group: serial pointer-graph work,
// SURVEY 2 "NodeManager ... OUT OF SCOPE for the device").  crass_build_outputs() of include/crass_hip.h.
//
//   f-4  NodeManager::addReadHolder / splitReadHolder / addCrisprNodes      src/crass/NodeManager.cpp:120-443
//        cleanGraph / clearBubbles (+ CrisprNode::setAttach's bookkeeping)  :689-945, CrisprNode.cpp:181-214
//        buildSpacerGraph / cleanSpacerGraph / removeSpacerBubbles          :1063-1291, SpacerInstance.cpp
//        splitIntoContigs / walkFromCross                                   :1293-1428
//        generateFlankers / getSpacerCountAndStats                          :2020-2068, :947-966
//        WorkHorse::buildGraph ... removeLowConfidenceNodeManagers          WorkHorse.cpp:454-577, 1642-1729
//   f-3  WorkHorse::outputResults / addDataToDOM / addMetadataToDOM         WorkHorse.cpp:1900-2249
//        NodeManager::addSpacersToDOM ... printAssemblyToDOM, dumpReads     NodeManager.cpp:1447-1753
//        printSpacerGraph / printSpacerKey, Rainbow                         :1789-2018, Rainbow.cpp
//        .crispr XML: tag / attribute names base.cpp:72-121, root <crispr version="1.1"> crassDefines.h:105-106,
//        text layout = Xerces-C 3.1.1 DOMLSSerializer with format-pretty-print (writer.cpp printDOMToFile)
//
// Data layout instead of the reference's heap objects: nodes and spacers live in vectors and name each other by index;
// a node's four edge lists are small sorted vectors of (partner node id, active) — sorted by node id, i.e. creation
// order, where the reference's std::map<CrisprNode*, bool> is sorted by heap address (its own output is address
// dependent there; oracle/crass_graph.py states the same choice); spacers are visited through one array sorted by the
// reference's 32-bit SpacerKey (its wrap-around included).  No Xerces: the XML is written as text.
#include "../../include/crass_hip.h"
#include <mutex>
#include <thread>
#include "merge.h"            // host_parallel_for

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

using std::string;

enum { EF = 0, EB = 1, EJF = 2, EJB = 3 };                  // CN_EDGE_FORWARD, _BACKWARD, _JUMPING_F, _JUMPING_B
enum { D_REVERSE = 0, D_FORWARD = 1 };                      // SI_EdgeDirection

inline uint32_t spacer_key(int a, int b)                    // makeSpacerKey (SpacerInstance.h:83-93), int overflow as on x86
{
    const uint32_t lo = (uint32_t)std::min(a, b), hi = (uint32_t)std::max(a, b);
    return lo * 10000000u + hi;
}
inline int bubble_key(int i, int j) { return (int)((uint32_t)i * 100000u + (uint32_t)j); }      // makeKey (NodeManager.h:88)

struct EdgeList {                                           // partner node id -> active, ascending id
    std::vector<std::pair<int, bool>> v;
    std::pair<int, bool> *find(int id)
    {
        auto it = std::lower_bound(v.begin(), v.end(), std::make_pair(id, false));
        return (it != v.end() && it->first == id) ? &*it : nullptr;
    }
    void set(int id, bool on)
    {
        auto it = std::lower_bound(v.begin(), v.end(), std::make_pair(id, false));
        if (it != v.end() && it->first == id) it->second = on;
        else v.insert(it, std::make_pair(id, on));
    }
};

struct Node {                                               // CrisprNode
    int id = 0, coverage = 1, rank[4] = {0, 0, 0, 0};
    bool attached = true, forward = true;
    EdgeList e[4];
    std::vector<int> headers;
    int total() const { return rank[0] + rank[1] + rank[2] + rank[3]; }
    int inner() const { return rank[EF] + rank[EB]; }
    int jumping() const { return rank[EJF] + rank[EJB]; }
};

struct SpEdge { int to; int d; };
struct Spacer {                                             // SpacerInstance
    int id = 0, leader = 0, last = 0, contig = 0;           // id: string token; leader / last: node ids
    unsigned count = 1;
    bool attached = false, flanker = false;
    std::vector<SpEdge> edges;
    int rank() const { return (int)edges.size(); }
};

struct ReadRef { const char *hdr, *com, *seq; uint32_t nh, nc, ns; const uint32_t *ss; uint32_t nss; int hst; };

struct Rainbow {                                            // Rainbow.cpp:47-208, type BLUE_RED
    double lb = 0, ub = 1, upper = 0, lower = 0, mult = 0, tick = 0, red_off = 0, blue_off = 0;
    int res = 10;
    static constexpr double PI = 3.1415927, DIV = 0.6666666666;
    Rainbow() { set_type(); set_limits(0, 1, 10); }
    void set_type() { red_off = DIV * PI; blue_off = 0; lower = 0; upper = DIV * PI; mult = (upper - lower) / (ub - lb); }
    void set_limits(double l, double u, int r) { lb = l; ub = u; res = r; mult = (upper - lower) / (ub - lb); tick = (ub - lb) / (double)(res - 1); }
    static string rgb(double v)
    {
        if (!(v == v) || std::isinf(v)) return "00";        // (int)NaN is INT_MIN on x86-64: "0 >= rgb"
        const int x = (int)v;
        if (x <= 0) return "00";
        char b[16];
        snprintf(b, sizeof b, x < 16 ? "0%x" : "%x", x);
        return b;
    }
    string colour(double value) const
    {
        if (res == -1) return "000000";
        if (value > ub || value < lb) return "000000";
        const double norm = std::round(value / tick) * tick;
        const double scaled = (norm - lb) * mult + lower;
        auto val = [](double x) { return (std::cos(x) + 0.5) * DIV; };
        return rgb(std::round(val(scaled - red_off) * 255)) + "00" + rgb(std::round(val(scaled - blue_off) * 255));
    }
};

struct Manager {                                            // NodeManager: one DR group
    string dr;
    int kmer = 7;
    // StringCheck: addString always makes a new token (first = 2); string -> latest token
    std::vector<string> tok_str;                            // token t -> tok_str[t - 2]
    std::unordered_map<string, int> tok_of;
    std::vector<int> node_of_tok;                           // token -> node index or -1
    std::vector<int> canon;                                 // token -> the FIRST token with the same string (headers may repeat)
    std::vector<Node> nodes;                                // creation order == ascending id
    std::vector<Spacer> spacers;                            // creation order
    std::unordered_map<uint32_t, int> sp_by_key;
    std::vector<std::pair<uint32_t, int>> sp_sorted;        // (key, spacer index) ascending key: the reference's map order
    std::vector<ReadRef> reads;
    int next_contig = 0;
    std::vector<size_t> stats;
    std::vector<int> flankers;
    Rainbow rainbow;
    string *out = nullptr;                                  // stdout text the reference prints on the way

    int add_string(const string &s)
    {
        tok_str.push_back(s);
        const int t = (int)tok_str.size() + 1;
        canon.resize(t + 1, 0);
        auto ins = tok_of.emplace(s, t);
        if (ins.second) canon[t] = t;
        else { canon[t] = canon[ins.first->second]; ins.first->second = t; }
        node_of_tok.resize(t + 1, -1);
        return t;
    }
    int get_token(const string &s) const { auto it = tok_of.find(s); return it == tok_of.end() ? 0 : it->second; }
    const string &str(int t) const { return tok_str[t - 2]; }
    Node &N(int id) { return nodes[node_of_tok[id]]; }

    // ---- the spacer cutter of ReadHolder (ReadHolder.cpp:813-952) ----
    struct Cutter {
        const ReadRef &r; int next = 0;
        explicit Cutter(const ReadRef &rr) : r(rr) {}
        bool get(string *o)
        {
            const uint32_t *ss = r.ss; const int n = (int)r.nss; const uint32_t L = r.ns;
            if (next > n - 1) return false;
            if (next == 0) {
                if (ss[0] != 0) { o->assign(r.seq, std::min<uint32_t>(ss[0], L)); next = 1; return true; }
                const uint32_t start = ss[1] + 1;
                if (start > L) throw 99;
                if (n > 2) o->assign(r.seq + start, ss[2] >= start ? std::min<uint32_t>(ss[2] - start, L - start) : L - start);
                else o->assign(r.seq + start, L - start);
                next = 3;
                return true;
            }
            const uint32_t v = ss[next];
            if (next == n - 1) {
                if ((uint64_t)v < (uint64_t)L - 1) { o->assign(r.seq + v + 1, L - v - 1); next += 2; return true; }      // (size_t compare)
                return false;
            }
            const uint32_t start = v + 1;
            if (start > L) throw 99;
            const uint32_t e = ss[next + 1];
            o->assign(r.seq + start, e >= start ? std::min<uint32_t>(e - start, L - start) : L - start);
            next += 2;
            return true;
        }
    };

    int node_for(const string &km, bool fwd)
    {
        int st = get_token(km);
        if (st == 0) {
            st = add_string(km);
            Node nd; nd.id = st; nd.forward = fwd;
            node_of_tok[st] = (int)nodes.size();
            nodes.push_back(std::move(nd));
        } else N(st).coverage++;
        return st;
    }
    void add_edge(int from, int to, int t)
    {
        Node &a = N(from);
        if (!a.e[t].find(to)) { a.e[t].set(to, true); a.rank[t]++; }
    }
    void link_prev(int prev, int first)
    {
        if (prev && !sp_by_key.count(spacer_key(first, prev))) { add_edge(prev, first, EJF); add_edge(first, prev, EJB); }
    }
    void add_nodes(int &prev, const string &ws, int hst)    // addCrisprNodes
    {
        if ((int)ws.size() < kmer) return;
        const int n1 = node_for(ws.substr(0, kmer), true);
        const int n2 = node_for(ws.substr(ws.size() - kmer), false);
        N(n1).headers.push_back(hst); N(n2).headers.push_back(hst);
        link_prev(prev, n1);
        const uint32_t key = spacer_key(n1, n2);
        auto it = sp_by_key.find(key);
        if (it == sp_by_key.end()) {
            int st = get_token(ws);
            if (!st) st = add_string(ws);
            Spacer s; s.id = st; s.leader = n1; s.last = n2;
            sp_by_key[key] = (int)spacers.size();
            spacers.push_back(std::move(s));
            add_edge(n1, n2, EF); add_edge(n2, n1, EB);
        } else spacers[it->second].count++;
        prev = n2;
    }
    void add_second(int &prev, const string &ws, int hst)
    {
        if ((int)ws.size() < kmer) return;
        const int n2 = node_for(ws.substr(ws.size() - kmer), false);
        N(n2).headers.push_back(hst);
        prev = n2;
    }
    void add_first(int &prev, const string &ws, int hst)
    {
        if ((int)ws.size() < kmer) return;
        const int n1 = node_for(ws.substr(0, kmer), true);
        N(n1).headers.push_back(hst);
        link_prev(prev, n1);
    }
    bool add_read(const ReadRef &r)                          // addReadHolder / splitReadHolder
    {
        const int hst = add_string(string(r.hdr, r.nh));
        Cutter c(r);
        string ws, tmp;
        int prev = 0;
        if (r.nss < 2 || !c.get(&ws)) return false;
        if (r.ss[0] == 0) add_nodes(prev, ws, hst); else add_second(prev, ws, hst);
        if (r.ns == r.ss[r.nss - 1] + 1) {
            while (c.get(&ws)) add_nodes(prev, ws, hst);
        } else {
            while (c.next < (int)r.nss - 1) { (void)c.get(&ws); add_nodes(prev, ws, hst); }
            if (c.get(&ws)) add_first(prev, ws, hst);
        }
        reads.push_back(r);
        reads.back().hst = hst;
        return true;
    }
    void sort_spacers()
    {
        sp_sorted.clear();
        for (auto &kv : sp_by_key) sp_sorted.push_back(kv);
        std::sort(sp_sorted.begin(), sp_sorted.end());
    }

    // ---- CrisprNode::setAttach with the reference's same-type bookkeeping (CrisprNode.cpp:181-214) ----
    void set_attach(int id, bool state)
    {
        for (int t = 0; t < 4; t++) {
            const size_t n = N(id).e[t].v.size();
            for (size_t q = 0; q < n; q++) {
                const int pid = N(id).e[t].v[q].first;
                if ((N(id).e[t].v[q].second != state) && N(pid).attached) {
                    N(pid).e[t].set(id, state);              // the partner's list of the SAME type
                    N(id).e[t].find(pid)->second = state;    // (pid == id: the insert above cannot have moved it, the key exists)
                    N(pid).rank[t] += state ? 1 : -1;
                    if (N(pid).total() == 0) N(pid).attached = false;
                }
            }
        }
        N(id).attached = state;
    }
    int discounted_coverage(int id)
    {
        Node &n = N(id);
        std::map<int, int> cm;
        for (int h : n.headers) cm[h] = 0;
        const int lists[2] = {n.forward ? EF : EJF, n.forward ? EJB : EB};
        for (int t : lists)
            for (auto &pe : n.e[t].v) {
                if (!pe.second) continue;
                for (int h : N(pe.first).headers) { auto it = cm.find(h); if (it != cm.end()) it->second++; }
            }
        int r = 0;
        for (auto &kv : cm) if (kv.second > 1) r++;
        return r;
    }
    void find_all(std::vector<int> &caps, std::vector<int> &other)
    {
        caps.clear(); other.clear();
        for (auto &n : nodes) if (n.attached) (n.total() == 1 ? caps : other).push_back(n.id);
    }
    int caps_at(bool forward, int q)                        // findCapsAt(.., isInner = true, doStrict = true, ..)
    {
        int caps = 0;
        if (!N(q).attached) return 0;
        for (auto &pe : N(q).e[forward ? EF : EB].v)
            if (pe.second) { if (N(pe.first).total() == 1) caps++; else return 0; }
        return caps;
    }
    bool clear_bubbles(int root, int t)
    {
        bool some = false;
        const int opp = t == EF ? EJF : EF;                 // (called with FORWARD and JUMPING_F only)
        std::map<int, int> bubble;
        // std::map iteration semantics: the next entry is the smallest key above the current one, so entries that set_attach
        // inserts behind the iterator while the loop runs are visited and those before it are not
        auto next_key = [](const EdgeList &l, int cur) {
            auto it = std::upper_bound(l.v.begin(), l.v.end(), std::make_pair(cur, true));
            return it == l.v.end() ? -1 : it->first;
        };
        for (int e = next_key(N(root).e[t], -1); e >= 0; e = next_key(N(root).e[t], e)) {
            if (!N(e).attached) continue;
            for (int e2 = next_key(N(e).e[opp], -1); e2 >= 0; e2 = next_key(N(e).e[opp], e2)) {
                if (!N(e2).attached) continue;
                const int key = bubble_key(root, e2);
                auto it = bubble.find(key);
                if (it == bubble.end()) bubble[key] = e;
                else {
                    const int first = it->second;
                    if (discounted_coverage(first) > discounted_coverage(e)) set_attach(e, false);
                    else { set_attach(first, false); bubble[key] = e; }
                    some = true;
                }
            }
        }
        return some;
    }
    int clean_graph()
    {
        bool some = true;
        std::vector<int> caps, other, detach;
        while (some) {
            some = false;
            std::vector<std::pair<int, int>> fork;          // (joining node, cap): the reference's multimap
            detach.clear();
            find_all(caps, other);
            for (int c : caps) {
                Node &n = N(c);
                if (n.inner() == 0) {
                    EdgeList &el = n.e[n.rank[EJF] != 0 ? EJF : EJB];
                    if (el.v.empty()) return 1;             // (the reference dereferences begin() of an empty map here)
                    if (N(el.v[0].first).total() != 2) detach.push_back(c);
                } else {
                    const bool use_f = n.rank[EF] != 0;
                    EdgeList &el = n.e[use_f ? EF : EB];
                    if (el.v.empty()) return 1;
                    const int j = el.v[0].first;
                    if (N(j).total() != 2) {
                        if (caps_at(!use_f, j) > 1) fork.push_back(std::make_pair(j, c));
                        else detach.push_back(c);
                    }
                }
            }
            std::stable_sort(fork.begin(), fork.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
            std::map<int, int> best;
            for (auto &f : fork) { auto it = best.find(f.first); if (it == best.end() || N(it->second).coverage < N(f.second).coverage) best[f.first] = f.second; }
            for (auto &f : fork) if (best[f.first] != f.second) detach.push_back(f.second);
            if (!detach.empty()) some = true;
            for (int d : detach) set_attach(d, false);
            find_all(caps, other);
            for (int o : other) {
                const int r = N(o).total();
                if (r == 2) { if (!(N(o).inner() && N(o).jumping())) { set_attach(o, false); some = true; } }
                else if (r == 0 || r == 1) {}
                else {
                    if (N(o).inner() != 1 && clear_bubbles(o, EF)) some = true;
                    if (N(o).jumping() != 1 && clear_bubbles(o, EJF)) some = true;
                }
            }
        }
        return 0;
    }

    // ---- spacer graph ----
    bool sp_attached(const Spacer &s)                       // SpacerInstance::isAttached incl. its message
    {
        if (N(s.leader).attached && (N(s.last).attached ^ s.attached)) *out += "Spacer " + std::to_string(s.id) + " has asynchronous attached state\n";
        return s.attached;
    }
    int build_spacer_graph()
    {
        sort_spacers();
        for (auto &ks : sp_sorted) {
            const int si = ks.second;
            if (N(spacers[si].last).attached && N(spacers[si].leader).attached) {
                spacers[si].attached = true;
                for (auto &qe : N(spacers[si].last).e[EJF].v) {
                    const int q = qe.first;
                    if (!(N(q).attached && N(q).forward)) continue;
                    for (auto &ee : N(q).e[EF].v) {
                        if (!N(ee.first).attached) continue;
                        auto it = sp_by_key.find(spacer_key(ee.first, q));
                        if (it == sp_by_key.end()) return 1;        // (the reference would insert a NULL spacer and crash)
                        const int nx = it->second;
                        if (nx != si) { spacers[si].edges.push_back({nx, D_FORWARD}); spacers[nx].edges.push_back({si, D_REVERSE}); }
                    }
                }
            } else spacers[si].attached = false;
        }
        return 0;
    }
    int detach_spacer(int si)                               // detachFromSpacerGraph
    {
        Spacer &s = spacers[si];
        if (s.rank() == 0) return 0;
        for (auto &e : s.edges) {
            auto &te = spacers[e.to].edges;
            auto it = std::find_if(te.begin(), te.end(), [si](const SpEdge &x) { return x.to == si; });
            if (it == te.end()) return 1;                   // (the reference leaves dangling edges behind here)
            te.erase(it);
        }
        s.edges.clear();
        return 0;
    }
    bool is_fur(const Spacer &s) const
    {
        if (s.rank() != 1) return false;
        for (auto &e : s.edges) if (spacers[e.to].rank() > 2) return true;
        return false;
    }
    static bool is_viable(const Spacer &s)
    {
        if (s.rank() < 2) return true;
        bool f = false, r = false;
        for (auto &e : s.edges) { if (e.d == D_REVERSE) r = true; else f = true; }
        return f && r;
    }
    bool sp_find(const Spacer &s, int other) const { for (auto &e : s.edges) if (e.to == other) return true; return false; }
    int remove_spacer_bubbles()
    {
        std::map<uint32_t, int> bubble;
        std::vector<int> detach;
        for (auto &ks : sp_sorted) {
            const int ci = ks.second;
            if (!sp_attached(spacers[ci])) continue;
            if (spacers[ci].rank() < 2) continue;
            std::vector<int> rs, fs;
            for (auto &e : spacers[ci].edges) (e.d == D_REVERSE ? rs : fs).push_back(e.to);
            for (int r : rs)
                for (int f : fs) {
                    const uint32_t key = spacer_key(spacers[r].id, spacers[f].id);
                    auto it = bubble.find(key);
                    if (it == bubble.end()) { bubble[key] = ci; continue; }
                    const int st = it->second;
                    if (sp_find(spacers[r], ci) && sp_find(spacers[r], st)) continue;
                    *out += "Coverage test: " + std::to_string(spacers[st].id) + " : " + std::to_string(spacers[ci].id) + "\n";
                    if (spacers[st].count < spacers[ci].count) { detach.push_back(st); it->second = ci; }
                    else if (spacers[ci].count < spacers[st].count) detach.push_back(ci);
                    else if (spacers[st].rank() < spacers[ci].rank()) { detach.push_back(st); it->second = ci; }
                    else detach.push_back(ci);
                }
        }
        for (int d : detach) if (detach_spacer(d)) return 1;
        return 0;
    }
    int clean_spacer_graph()
    {
        bool cleaned = true;
        while (cleaned) {
            cleaned = false;
            for (auto &ks : sp_sorted) if (sp_attached(spacers[ks.second]) && is_fur(spacers[ks.second])) { if (detach_spacer(ks.second)) return 1; cleaned = true; }
            for (auto &ks : sp_sorted) if (sp_attached(spacers[ks.second]) && !is_viable(spacers[ks.second])) { if (detach_spacer(ks.second)) return 1; cleaned = true; }
            if (remove_spacer_bubbles()) return 1;
        }
        return 0;
    }

    // ---- contigs ----
    struct Walk { int first = -1, second = -1, want = 0; };
    bool edge_from_cap(Walk &w, int ci)
    {
        Spacer &c = spacers[ci];
        if (c.rank() != 1) return false;
        for (auto &e : c.edges) {
            if (!sp_attached(spacers[e.to])) return false;
            if (spacers[e.to].contig == 0) { w.second = e.to; w.first = ci; w.want = e.d; }
            else { c.contig = spacers[e.to].contig; return false; }
        }
        return !(w.first < 0 || w.second < 0);
    }
    bool edge_from_cross(Walk &w, int ci)
    {
        Spacer &c = spacers[ci];
        if (c.rank() != 2) return false;
        for (auto &e : c.edges) {
            if (!sp_attached(spacers[e.to])) return false;
            if (spacers[e.to].contig == 0) { w.second = e.to; w.first = ci; w.want = e.d; return true; }
        }
        return !(w.first < 0 || w.second < 0);
    }
    bool step(Walk &w, int &prev)
    {
        if (spacers[w.second].rank() != 2) return false;
        for (auto &e : spacers[w.second].edges)
            if (sp_attached(spacers[e.to]) && e.d == w.want && spacers[e.to].id != spacers[w.first].id && spacers[e.to].contig == 0) {
                prev = w.first; w.first = w.second; w.second = e.to;
                return true;
            }
        return false;
    }
    int split_into_contigs()
    {
        Walk w;
        std::vector<int> start, cross;
        for (auto &ks : sp_sorted) if (sp_attached(spacers[ks.second]) && spacers[ks.second].rank() == 1) start.push_back(ks.second);
        for (int cap : start) {
            std::vector<int> cur;
            next_contig++;
            if (edge_from_cap(w, cap)) {
                int prev = -1;
                do { if (prev >= 0) cur.push_back(prev); } while (step(w, prev));
                cur.push_back(w.first);
                if (spacers[w.second].rank() == 1) cur.push_back(w.second); else cross.push_back(w.second);
                for (int s : cur) spacers[s].contig = next_contig;
            }
        }
        next_contig++;
        Walk w2;                                            // walkFromCross: a fresh WalkingManager
        for (size_t i = 0; i < cross.size(); i++) {
            const int c = cross[i];
            spacers[c].contig = next_contig++;
            const std::vector<SpEdge> edges = spacers[c].edges;
            for (auto &e : edges) {
                if (!(sp_attached(spacers[e.to]) && spacers[e.to].contig == 0)) continue;
                if (edge_from_cross(w2, e.to)) {
                    std::vector<int> cur;
                    int prev = -1;
                    do { if (prev >= 0) cur.push_back(prev); } while (step(w2, prev));
                    if (spacers[w2.second].rank() == 1 && sp_attached(spacers[w2.second])) cur.push_back(w2.second);
                    else if (spacers[w2.second].contig == 0 && sp_attached(spacers[w2.second])) { cur.push_back(w2.first); cross.push_back(w2.second); }
                    for (int s : cur) spacers[s].contig = next_contig;
                    next_contig++;
                } else cross.push_back(e.to);
            }
        }
        return 0;
    }

    // ---- stats / flankers ----
    int count_and_stats(bool show_detached = false, bool exclude_flankers = true)
    {
        int n = 0;
        for (auto &ks : sp_sorted) {
            Spacer &s = spacers[ks.second];
            if (show_detached || sp_attached(s)) {
                if (exclude_flankers && s.flanker) continue;
                stats.push_back(str(s.id).size());
                n++;
            }
        }
        return n;
    }
    size_t mean() const { size_t a = 0; for (size_t v : stats) a += v; return a / stats.size(); }
    double stdev() const
    {
        const double avg = (double)mean();
        double acc = 0;
        for (size_t v : stats) { const double d = (double)v - avg; acc += d * d; }
        return std::sqrt(acc / (double)stats.size());
    }
    void generate_flankers()
    {
        const int n = count_and_stats();
        if (n >= 3) {
            const double sd = stdev();
            const int mn = (int)mean();
            const int lower = (int)(mn - (sd * 1.5)), upper = (int)(mn + (sd * 1.5));
            if (sd > 1)
                for (auto &ks : sp_sorted) {
                    Spacer &s = spacers[ks.second];
                    if (N(s.leader).attached && N(s.last).attached) {
                        const int len = (int)str(s.id).size();
                        if (len > upper || len < lower) { s.flanker = true; flankers.push_back(ks.second); }
                    }
                }
        }
        stats.clear();
    }

    // ---- the text outputs ----
    string label(const Spacer &s, bool long_desc) const
    {
        string l = s.flanker ? "fl_" : "sp_";
        l += std::to_string(s.id);
        if (long_desc) l += "_" + str(s.id);
        l += "_" + std::to_string(s.count) + "_C" + std::to_string(s.contig);
        return l;
    }
    bool spacer_graph_text(const string &title, bool long_desc, bool show_singles, string *text)
    {
        double mx = 0, mn = 10000000;
        for (auto &ks : sp_sorted) { const double c = (double)spacers[ks.second].count; if (c > mx) mx = c; else if (c < mn) mn = c; }
        rainbow.set_type();
        rainbow.set_limits(mn, mx, (int)(mx - mn) + 1);
        string t = "digraph " + title + " {\n";
        std::vector<int> sel;
        for (auto &ks : sp_sorted) {
            Spacer &s = spacers[ks.second];
            if (sp_attached(s) && (show_singles || s.rank() != 0)) {
                sel.push_back(ks.second);
                const string col = rainbow.colour((double)s.count);
                t += "\t\t" + label(s, long_desc) + " [ color = \"#" + col + "\", fillcolor=\"#" + col + "\", style= filled, shape=" + (s.flanker ? "diamond" : "circle") + "];\n";
            }
        }
        if (sel.empty()) return false;
        for (int si : sel)
            for (auto &e : spacers[si].edges) {
                Spacer &tg = spacers[e.to];
                if (sp_attached(tg) && e.d == D_FORWARD && (show_singles || tg.rank() != 0)) t += "\t\t" + label(spacers[si], long_desc) + " -> " + label(tg, long_desc) + " [ len=2 ];\n";
            }
        *text = t + "\n}\n";
        return true;
    }
    string spacer_key_text(int cluster, const string &name) const
    {
        string t = "\tsubgraph cluster_" + std::to_string(cluster) + "\t{\n\t\t\"" + name + "\" [ fillcolor = \"white\" shape = \"record\" label =<<table border=\"0\" cellborder=\"0\" "
                   "cellpadding=\"0\" bgcolor=\"white\"><tr><td>" + name + "</td></tr>";
        const double ul = rainbow.ub, ll = rainbow.lb;
        double step = (ul - ll) / 9;
        if (step < 1) step = 1;
        for (double i = ll; i <= ul; i += step) {
            const int ts = (int)i;
            t += "<tr><td bgcolor=\"#" + rainbow.colour(ts) + "\" align=\"center\" colspan=\"2\"><font color=\"white\">" + std::to_string(ts) + "</font></td></tr>";
        }
        return t + "</table>> ];\n\t}\n";
    }
    string dump_reads_text()                                // dumpReads(.., showDetached = true) + ReadHolder::print
    {
        // reads_set holds header STRINGS in the reference; here: one mark per distinct string (its first token)
        std::vector<char> mark(canon.size(), 0);
        for (auto &ks : sp_sorted) {
            Spacer &s = spacers[ks.second];
            for (int nd : {s.leader, s.last}) for (int h : N(nd).headers) mark[canon[h]] = 1;
        }
        size_t bytes = 0;
        for (auto &r : reads) if (mark[canon[r.hst]]) bytes += (size_t)r.nh + r.nc + r.ns + 4;
        string o;
        o.reserve(bytes);
        for (auto &r : reads) {
            if (!mark[canon[r.hst]]) continue;
            o += ">"; o.append(r.hdr, r.nh);
            if (r.nc) { o += " "; o.append(r.com, r.nc); }
            o += "\n"; o.append(r.seq, r.ns); o += "\n";
        }
        return o;
    }
};

// ---- XML as Xerces-C's DOMLSSerializer (format-pretty-print) writes it ----
struct Xml {
    string tag, text;
    bool has_text = false;
    std::vector<std::pair<string, string>> attrs;
    std::vector<std::unique_ptr<Xml>> kids;
    string raw;                                             // further children, already laid out as text (the long leaf lists)
    explicit Xml(const string &t) : tag(t) {}
    // one empty-element child <tag a1="v1" [a2="v2"]/> at `level` as text; attribute names must be given in name order
    static void leaf(string &o, int level, const char *tag, const char *a1, const string &v1, const char *a2 = nullptr, const string *v2 = nullptr)
    {
        o += "\n"; o.append((size_t)level * 2, ' '); o += "<"; o += tag;
        o += " "; o += a1; o += "=\""; esc(o, v1, true); o += "\"";
        if (a2) { o += " "; o += a2; o += "=\""; esc(o, *v2, true); o += "\""; }
        o += "/>";
    }
    Xml *add(const string &t) { kids.emplace_back(new Xml(t)); return kids.back().get(); }
    Xml *attr(const string &k, const string &v) { attrs.push_back(std::make_pair(k, v)); return this; }
    Xml *txt(const string &t) { text = t; has_text = true; return this; }
    static void esc(string &o, const string &s, bool attr)
    {
        for (unsigned char ch : s) {
            if (ch == '&') o += "&amp;";
            else if (ch == '<') o += "&lt;";
            else if (attr && ch == '"') o += "&quot;";
            else if (!attr && ch == '>') o += "&gt;";
            else if (attr && (ch == 9 || ch == 10 || ch == 13)) { char b[8]; snprintf(b, sizeof b, "&#x%X;", ch); o += b; }
            else o += (char)ch;
        }
    }
    void write(string &o, int level)
    {
        if (level == 1) o += "\n";                          // format-pretty-print-1st-level
        o += "\n"; o.append((size_t)level * 2, ' '); o += "<"; o += tag;
        std::sort(attrs.begin(), attrs.end());              // DOMAttrMapImpl keeps attributes sorted by name
        for (auto &a : attrs) { o += " "; o += a.first; o += "=\""; esc(o, a.second, true); o += "\""; }
        if (has_text) { o += ">"; esc(o, text, false); o += "</"; o += tag; o += ">"; }
        else if (!kids.empty() || !raw.empty()) {
            o += ">";
            for (auto &k : kids) k->write(o, level + 1);
            o += raw;
            if (level == 0) o += "\n";
            o += "\n"; o.append((size_t)level * 2, ' '); o += "</"; o += tag; o += ">";
        } else o += "/>";
    }
};

} // namespace

// fn(i) for i in [0, n) on up to `threads` threads of its own (the caller included): the output stage's copies and writes
static void team_for(size_t n, unsigned threads, const std::function<void(size_t)> &fn)
{
    threads = (unsigned)std::min<size_t>(threads, n);
    if (threads <= 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
    std::atomic<size_t> next{0};
    auto work = [&]() { for (;;) { const size_t i = next.fetch_add(1, std::memory_order_relaxed); if (i >= n) break; fn(i); } };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < threads; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
}

struct crass_outputs {
    std::vector<string> names, data;
    // a file put together from the groups' texts (crass.crispr: 100 MB for a 50 M-read job) is a buffer of its own, filled by
    // several threads at once — a std::string of that size is zero-filled, then filled, by one; data[i] is empty for such a file
    struct Blob { char *p = nullptr; size_t n = 0; };
    std::vector<Blob> blob;                                 // [file]: p != nullptr: the file's bytes
    std::vector<const char *> name_p, data_p;
    std::vector<uint64_t> sizes;
    std::vector<int32_t> kept;
    string out;
    std::vector<std::unique_ptr<Manager>> dead;             // the groups' graphs: taken apart with the outputs
    unsigned dead_threads = 1;
    ~crass_outputs()
    {
        if (dead_threads > 1) crass::host_parallel_for(dead.size(), dead_threads, [&](size_t i) { dead[i].reset(); });
        dead.clear();
        for (Blob &b : blob) free(b.p);
    }
};

extern "C" {

int crass_build_outputs(const crass_graph_input *in, const crass_output_opts *op, crass_outputs **res)
{
    if (!in || !op || !res) return CRASS_ERR_INVALID_ARG;
    *res = nullptr;
    if (in->n_groups && (!in->gid || !in->dr_chars || !in->dr_off || !in->grp_rec_off)) return CRASS_ERR_INVALID_ARG;
    if (in->n_rec && (!in->hdr_chars || !in->hdr_off || !in->seq_chars || !in->seq_off || !in->rec_nss || !in->rec_ss_off || !in->ss_pool)) return CRASS_ERR_INVALID_ARG;
    if (in->n_rec && in->com_chars && !in->com_off) return CRASS_ERR_INVALID_ARG;       // comments given without their offsets
    // (contract, include/crass_hip.h: ss_pool holds rec_ss_off[k] + rec_nss[k] entries for every record k — the pool's length is not
    // part of the ABI, so it cannot be checked here; start/stop VALUES are checked against the read's length where they are used)
    std::unique_ptr<crass_outputs> R(new crass_outputs());
    const bool timing = getenv("CRASS_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[crass_timing] outputs: %-40s %.4f s\n", what, t - t_prev); t_prev = t; } };
    const string outdir = op->out_dir ? op->out_dir : "./", stamp = op->timestamp ? op->timestamp : "", cwd = op->cwd ? op->cwd : "";
    const string package = "crass", version = "1.0.1";     // PACKAGE_NAME / PACKAGE_VERSION (configure.ac:5)
    const int kmer = op->node_kmer > 0 ? op->node_kmer : 7, cov_cutoff = op->cov_cutoff > 0 ? op->cov_cutoff : 3;
    // buildGraph (WorkHorse.cpp:454-505): mDRs is keyed by the true DR string
    std::vector<std::unique_ptr<Manager>> own;
    std::map<string, Manager *> by_dr;
    std::vector<Manager *> of_group(in->n_groups, nullptr);
    // One NodeManager per DR group, and nothing of one group's graph depends on another's: the stages run over the groups on the
    // host pool (64 groups of ~1 600 reads at 10 M reads: 0.16 s of a 0.9 s crass-hip run on one thread).  What the reference
    // prints on the way is collected per stage and group and put together in the order one thread would have produced it.
    const uint32_t ng = in->n_groups;
    for (uint32_t g = 0; g < ng; g++) {
        if (g && in->gid[g] <= in->gid[g - 1]) return CRASS_ERR_INVALID_ARG;
        if (in->grp_rec_off[g + 1] > in->n_rec || in->grp_rec_off[g + 1] < in->grp_rec_off[g]) return CRASS_ERR_INVALID_ARG;
        own.emplace_back(new Manager());
        Manager *m = own.back().get();
        m->dr.assign(in->dr_chars + in->dr_off[g], (size_t)(in->dr_off[g + 1] - in->dr_off[g]));
        m->kmer = kmer;
        by_dr[m->dr] = m;
    }
    for (uint32_t g = 0; g < ng; g++) of_group[g] = by_dr[own[g]->dr];
    // (two groups with one DR string share the LAST one's manager, as the reference's map does: then one thread, in order)
    const unsigned threads = by_dr.size() == ng ? 32u : 1u;
    std::vector<string> logs;                                // per stage and group / manager
    std::vector<int> bad;
    auto stage = [&](size_t n, const std::function<int(size_t, string *)> &fn) -> int {
        logs.assign(n, string()); bad.assign(n, 0);
        if (threads > 1) crass::host_parallel_for(n, threads, [&](size_t i) { bad[i] = fn(i, &logs[i]); });
        else for (size_t i = 0; i < n; i++) bad[i] = fn(i, &logs[i]);
        for (size_t i = 0; i < n; i++) { R->out += logs[i]; if (bad[i]) return bad[i]; }
        return 0;
    };
    {
        const int st = stage(ng, [&](size_t g, string *lg) -> int {
            Manager *m = own[g].get();
            m->out = lg;
            try {
                for (uint64_t k = in->grp_rec_off[g]; k < in->grp_rec_off[g + 1]; k++) {
                    ReadRef r;
                    r.hdr = in->hdr_chars + in->hdr_off[k]; r.nh = (uint32_t)(in->hdr_off[k + 1] - in->hdr_off[k]);
                    r.com = in->com_chars ? in->com_chars + in->com_off[k] : nullptr; r.nc = in->com_chars ? (uint32_t)(in->com_off[k + 1] - in->com_off[k]) : 0;
                    r.seq = in->seq_chars + in->seq_off[k]; r.ns = (uint32_t)(in->seq_off[k + 1] - in->seq_off[k]);
                    r.ss = in->ss_pool + in->rec_ss_off[k]; r.nss = in->rec_nss[k];
                    (void)m->add_read(r);
                }
            } catch (int) {
                return CRASS_ERR_SEARCH_FATAL;               // substring_exception: the reference exit(99)s (NodeManager.cpp:216-219)
            }
            return 0;
        });
        if (st) return st;
    }
    lap("reads -> nodes and spacers");
    std::vector<bool> alive(ng, true);
    { const int st = stage(ng, [&](size_t g, string *lg) -> int { of_group[g]->out = lg; return of_group[g]->clean_graph() ? CRASS_ERR_SEARCH_FATAL : 0; }); if (st) return st; }
    lap("cleanGraph");
    std::vector<Manager *> in_dr_order;                      // makeSpacerGraphs .. splitIntoContigs walk mDRs
    for (auto &kv : by_dr) in_dr_order.push_back(kv.second);
    { const int st = stage(in_dr_order.size(), [&](size_t i, string *lg) -> int { in_dr_order[i]->out = lg; return in_dr_order[i]->build_spacer_graph() ? CRASS_ERR_SEARCH_FATAL : 0; }); if (st) return st; }
    { const int st = stage(in_dr_order.size(), [&](size_t i, string *lg) -> int { in_dr_order[i]->out = lg; return in_dr_order[i]->clean_spacer_graph() ? CRASS_ERR_SEARCH_FATAL : 0; }); if (st) return st; }
    (void)stage(in_dr_order.size(), [&](size_t i, string *lg) -> int { in_dr_order[i]->out = lg; in_dr_order[i]->split_into_contigs(); return 0; });
    (void)stage(ng, [&](size_t g, string *lg) -> int { of_group[g]->out = lg; of_group[g]->generate_flankers(); return 0; });
    {
        std::vector<uint8_t> keep(ng, 1);
        (void)stage(ng, [&](size_t g, string *lg) -> int {   // removeLowConfidenceNodeManagers (WorkHorse.cpp:544-573)
            Manager *m = of_group[g];
            m->out = lg;
            if (m->count_and_stats(false) < cov_cutoff) keep[g] = 0;
            else if (m->stdev() > 6.0) keep[g] = 0;          // CRASS_DEF_STDEV_SPACER_LENGTH
            return 0;
        });
        for (uint32_t g = 0; g < ng; g++) alive[g] = keep[g] != 0;
    }
    for (auto &mp : own) mp->out = &R->out;                  // (from here on: one thread again where anything is printed)
    lap("spacer graphs, contigs, flankers");
    // outputResults (WorkHorse.cpp:1900-2038)
    auto put = [&](const string &name, string &&data) { R->names.push_back(name); R->data.push_back(std::move(data)); R->blob.emplace_back(); };      // (a group's read dump is tens of MB: moved, not copied)
    const string name_prefix = outdir + package + ".crispr";
    string keys = "digraph Keys {\n";
    Xml root("crispr");
    root.attr("version", "1.1");
    // per group: the spacer graph's text decides whether the group is kept (phase A); the kept groups are numbered in order
    // (the key graph's cluster index); then the key text, the read dump and the group's XML subtree — as text — per group
    // (phase B); the pieces are put together in group order
    std::vector<string> gv_txt(ng), key_txt(ng), fa_txt(ng), xml_txt(ng);
    {
        std::vector<uint8_t> ok(ng, 0);
        (void)stage(ng, [&](size_t g, string *lg) -> int {
            if (!alive[g]) return 0;
            Manager *m = of_group[g];
            m->out = lg;
            ok[g] = m->spacer_graph_text(m->dr, op->long_description != 0, op->show_singles != 0, &gv_txt[g]) ? 1 : 0;
            return 0;
        });
        for (uint32_t g = 0; g < ng; g++) if (alive[g] && !ok[g]) alive[g] = false;
    }
    std::mutex part_mu;
    double part_s[5] = {0, 0, 0, 0, 0};
    std::vector<int> cluster_of(ng, -1);
    { int cluster = 0; for (uint32_t g = 0; g < ng; g++) if (alive[g]) cluster_of[g] = cluster++; }
    (void)stage(ng, [&](size_t g, string *lg) -> int {
        if (!alive[g]) return 0;
        Manager *m = of_group[g];
        m->out = lg;
        const string gid = std::to_string(in->gid[g]);
        const string gv_name = "Spacers_" + gid + "_" + m->dr + "_spacers.gv", fa_name = "Group_" + gid + "_" + m->dr + ".fa";
        const double tp0 = timing ? now() : 0;
        key_txt[g] = m->spacer_key_text(cluster_of[g], name_prefix + gid);
        const double tp1 = timing ? now() : 0;
        fa_txt[g] = m->dump_reads_text();
        const double tp2 = timing ? now() : 0;
        Xml grp_node("group");
        Xml *grp = grp_node.attr("gid", "G" + gid)->attr("drseq", m->dr);
        // <data> (WorkHorse.cpp:2040-2088)
        Xml *data = grp->add("data");
        Xml *sources = data->add("sources"), *drs = data->add("drs"), *sps = data->add("spacers");
        Xml *fls = m->flankers.empty() ? nullptr : data->add("flankers");
        drs->add("dr")->attr("seq", m->dr)->attr("drid", "DR1");
        std::vector<char> in_all(m->tok_str.size() + 2, 0);   // all_sources as marks (tokens are dense)
        std::vector<int> toks;
        // a <spacer> / <flanker> element (level 4) with its <source soid=..> children (level 5, ascending token), written as text
        // the way Xml::write lays such a node out — a group has ten thousand of them, and as nodes each was a dozen allocations;
        // attribute names are given in name order
        auto spacer_text = [&](string &o, const char *tag, std::initializer_list<std::pair<const char *, string>> attrs, const Spacer &s) {
            o += "\n"; o.append(8, ' '); o += "<"; o += tag;
            for (auto &a : attrs) { o += " "; o += a.first; o += "=\""; Xml::esc(o, a.second, true); o += "\""; }
            toks.clear();
            for (int nd : {s.leader, s.last}) for (int h : m->N(nd).headers) toks.push_back(h);
            std::sort(toks.begin(), toks.end());
            toks.erase(std::unique(toks.begin(), toks.end()), toks.end());
            if (toks.empty()) { o += "/>"; return; }
            o += ">";
            for (int t : toks) { Xml::leaf(o, 5, "source", "soid", "SO" + std::to_string(t)); in_all[t] = 1; }
            o += "\n"; o.append(8, ' '); o += "</"; o += tag; o += ">";
        };
        for (auto &ks : m->sp_sorted) {
            const Spacer &s = m->spacers[ks.second];
            if (m->N(s.leader).attached && m->N(s.last).attached && !s.flanker)
                spacer_text(sps->raw, "spacer", {{"cov", std::to_string(s.count)}, {"seq", m->str(s.id)}, {"spid", "SP" + std::to_string(s.id)}}, s);
        }
        if (fls)
            for (int fi : m->flankers) {
                const Spacer &s = m->spacers[fi];
                if (m->N(s.leader).attached && m->N(s.last).attached) spacer_text(fls->raw, "flanker", {{"flid", "FL" + std::to_string(s.id)}, {"seq", m->str(s.id)}}, s);
            }
        for (int t = 2; t < (int)in_all.size(); t++)
            if (in_all[t]) { const string so = "SO" + std::to_string(t); Xml::leaf(sources->raw, 4, "source", "accession", m->str(t), "soid", &so); }
        // <metadata> (WorkHorse.cpp:2090-2249)
        Xml *meta = grp->add("metadata");
        Xml *prog = meta->add("program");
        prog->add("name")->txt(package); prog->add("version")->txt(version); prog->add("command")->txt(op->command_line ? op->command_line : "");
        meta->add("notes")->txt("Run on " + stamp);
        const string absdir = cwd + "/";
        if (!op->log_to_screen) meta->add("file")->attr("type", "log")->attr("url", absdir + outdir + package + "." + stamp + ".log");
        meta->add("file")->attr("type", "data")->attr("url", absdir + outdir + gv_name);
        meta->add("file")->attr("type", "sequence")->attr("url", absdir + outdir + fa_name);
        // <assembly> (NodeManager.cpp:1560-1706)
        Xml *asmb = grp->add("assembly");
        // (the spacers of every contig, in sp_sorted's order: one pass — every contig looking through every spacer is quadratic,
        // and a metagenome's group has thousands of one-spacer contigs)
        std::vector<std::vector<int>> of_contig((size_t)std::max(0, m->next_contig) + 1);
        for (auto &ks : m->sp_sorted) {
            const Spacer &s = m->spacers[ks.second];
            if (s.contig >= 1 && s.contig <= m->next_contig && m->sp_attached(s)) of_contig[(size_t)s.contig].push_back(ks.second);
        }
        const double tp3 = timing ? now() : 0;
        for (int cn = 1; cn <= m->next_contig; cn++) {
            Xml *ce = asmb->add("contig")->attr("cid", "C" + std::to_string(cn));
            for (int si : of_contig[(size_t)cn]) {
                Spacer &s = m->spacers[si];
                const string pre = s.flanker ? "FL" : "SP";
                Xml *cs = ce->add("cspacer")->attr("spid", pre + std::to_string(s.id));
                std::unique_ptr<Xml> part[4];                // bspacers, fspacers, bflankers, fflankers
                for (auto &e : s.edges) {
                    Spacer &tg = m->spacers[e.to];
                    if (!m->sp_attached(tg)) continue;
                    const string eid = pre + std::to_string(tg.id);
                    const bool fw = e.d == D_FORWARD;
                    const int slot = tg.flanker ? (fw ? 3 : 2) : (fw ? 1 : 0);
                    static const char *outer[4] = {"bspacers", "fspacers", "bflankers", "fflankers"}, *inner[4] = {"bs", "fs", "bf", "ff"};
                    if (!part[slot]) part[slot].reset(new Xml(outer[slot]));
                    Xml *x = part[slot]->add(inner[slot]);
                    if (tg.flanker) x->attr("flid", eid)->attr("drconf", "0")->attr("directjoin", "0");
                    else x->attr("drid", "DR1")->attr("drconf", "0")->attr("spid", eid);
                }
                for (auto &p : part) if (p) cs->kids.push_back(std::move(p));
            }
        }
        const double tp4 = timing ? now() : 0;
        grp_node.write(xml_txt[g], 1);
        if (timing) { const double tp5 = now(); std::lock_guard<std::mutex> lk(part_mu); part_s[0] += tp1 - tp0; part_s[1] += tp2 - tp1; part_s[2] += tp3 - tp2; part_s[3] += tp4 - tp3; part_s[4] += tp5 - tp4; }
        return 0;
    });
    if (timing) fprintf(stderr, "[crass_timing] outputs: per-group parts, summed over the groups: key text %.3f s, read dump %.3f s, <data> %.3f s, <assembly> %.3f s, XML text %.3f s\n",
                        part_s[0], part_s[1], part_s[2], part_s[3], part_s[4]);
    for (auto &mp : own) mp->out = &R->out;
    for (uint32_t g = 0; g < ng; g++) {
        if (!alive[g]) continue;
        const Manager *m = of_group[g];
        const string gid = std::to_string(in->gid[g]);
        put("Spacers_" + gid + "_" + m->dr + "_spacers.gv", std::move(gv_txt[g]));
        keys += key_txt[g];
        put("Group_" + gid + "_" + m->dr + ".fa", std::move(fa_txt[g]));
        R->kept.push_back(in->gid[g]);
    }
    lap("group files + XML tree");
    R->out += "[" + package + "_graphBuilder]: " + std::to_string(R->kept.size()) + " CRISPRs found!\n";
    // the document: <crispr>'s children are the groups' subtrees, already laid out as text — put together once, in a string of
    // the final size (a 50 M-read job's .crispr is 100 MB: through Xml::write it was copied three times into growing strings)
    string xml = "<?xml version=\"1.0\" encoding=\"ISO8859-1\" standalone=\"no\" ?>";
    {
        size_t total = 0;
        for (uint32_t g = 0; g < ng; g++) if (alive[g]) total += xml_txt[g].size();
        if (total == 0) { root.write(xml, 0); xml += "\n"; put(package + ".crispr", std::move(xml)); }
        else {
            string open_tag, close_tag;
            root.raw = "\x01";                              // (a placeholder child: the opening and closing text around it)
            root.write(open_tag, 0);
            const size_t cut = open_tag.find('\x01');
            close_tag = open_tag.substr(cut + 1);
            open_tag.resize(cut);
            close_tag += "\n";
            // the pieces' places, then every piece copied by whichever thread gets to it (the first touch of the buffer's pages
            // goes side by side too)
            std::vector<std::pair<const string *, size_t>> piece;
            size_t at = 0;
            auto add = [&](const string *t) { piece.emplace_back(t, at); at += t->size(); };
            add(&xml); add(&open_tag);
            for (uint32_t g = 0; g < ng; g++) if (alive[g]) add(&xml_txt[g]);
            add(&close_tag);
            crass_outputs::Blob bl;
            bl.n = at;
            bl.p = (char *)malloc(at ? at : 1);
            if (!bl.p) return CRASS_ERR_OOM;
            team_for(piece.size(), 8, [&](size_t i) {
                string *t = const_cast<string *>(piece[i].first);
                if (!t->empty()) memcpy(bl.p + piece[i].second, t->data(), t->size());
                string().swap(*t);                          // (a group's text goes back as soon as it has been copied, by the thread that copied it)
            });
            put(package + ".crispr", string());
            R->blob.back() = bl;
        }
    }
    put(package + "." + stamp + ".keys.gv", keys + "\n}\n");
    lap("XML text");
    for (size_t i = 0; i < R->names.size(); i++) {
        const bool bl = R->blob[i].p != nullptr;
        R->name_p.push_back(R->names[i].c_str()); R->data_p.push_back(bl ? R->blob[i].p : R->data[i].data()); R->sizes.push_back(bl ? R->blob[i].n : R->data[i].size());
    }
    // the managers are taken apart when the outputs are freed (crass_outputs_free: on the host pool, as they were built — one thread
    // freeing 50 graphs of 10 k reads each, node maps and spacer strings, was 0.4 s of a 50 M-read run).  Not here beside the
    // caller's crass_outputs_write: sixteen threads of tear-down and the writers together ran into the box's CPU quota, and a
    // process that ends behind its outputs never needs it
    R->dead.swap(own);
    R->dead_threads = threads;
    lap("tear-down handed on");
    *res = R.release();
    return CRASS_OK;
}

int crass_outputs_get(const crass_outputs *o, crass_outputs_view *v)
{
    if (!o || !v) return CRASS_ERR_INVALID_ARG;
    v->n_files = (uint32_t)o->names.size(); v->name = o->name_p.data(); v->data = o->data_p.data(); v->size = o->sizes.data();
    v->n_groups_kept = (uint32_t)o->kept.size(); v->kept_gid = o->kept.data(); v->stdout_text = o->out.c_str();
    return CRASS_OK;
}

int crass_outputs_write(const crass_outputs *o, const char *dir)
{
    if (!o) return CRASS_ERR_INVALID_ARG;
    string d = dir ? dir : "./";
    if (!d.empty() && d[d.size() - 1] != '/') d += '/';
    // eight threads create and write a file each, the largest first (a hundred group files and crass.crispr, 400 MB, through one
    // thread's fopen / fwrite were 0.12 s of a 50 M-read run — most of it the files' creation, 1.7 ms each on the GPU boxes; slices of
    // ONE file from several threads meet on the file's lock and were slower than that)
    const size_t nf = o->names.size();
    std::vector<int> fd(nf, -1);
    int rc = CRASS_OK;
    const bool timing = getenv("CRASS_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tw0 = now();
    std::vector<size_t> order(nf);
    for (size_t i = 0; i < nf; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return o->sizes[x] > o->sizes[y]; });
    std::atomic<int> bad{0};
    const double tw1 = now();
    double t_big = 0, t_open_sum = 0, t_open_max = 0, t_wr_sum = 0;
    std::mutex tm_mu;
    if (rc == CRASS_OK)
        team_for(nf, 8, [&](size_t k) {
            const size_t i = order[k];
            const double tf0 = k == 0 ? now() : 0;
            const double to0 = timing ? now() : 0;
            fd[i] = open((d + o->names[i]).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);      // (1.7 ms per file on the GPU boxes: by its writer)
            if (timing) { const double dt = now() - to0; std::lock_guard<std::mutex> lk(tm_mu); t_open_sum += dt; t_open_max = std::max(t_open_max, dt); }
            if (fd[i] < 0) { bad.store(1); return; }
            const char *p = o->data_p[i];
            size_t left = o->sizes[i];
            while (left) {
                const ssize_t w = write(fd[i], p, left);
                if (w < 0 && errno == EINTR) continue;
                if (w <= 0) { bad.store(1); return; }
                p += w; left -= (size_t)w;
            }
            // (closed at once: a hundred descriptors held until the end grow the process's descriptor table, and in a process with
            // threads every doubling of that table waits for an RCU grace period — 30 .. 50 ms each, 0.1 s of this call)
            if (close(fd[i]) != 0) bad.store(1);
            fd[i] = -1;
            if (k == 0) t_big = now() - tf0;
            if (timing) { const double dt = now() - to0; std::lock_guard<std::mutex> lk(tm_mu); t_wr_sum += dt; }
        });
    const double tw2 = now();
    for (size_t i = 0; i < nf; i++) if (fd[i] >= 0 && close(fd[i]) != 0) rc = CRASS_ERR_IO;
    if (bad.load()) rc = CRASS_ERR_IO;
    if (timing) {
        uint64_t total = 0;
        for (size_t i = 0; i < nf; i++) total += o->sizes[i];
        fprintf(stderr, "[crass_timing] outputs: %zu files, %.1f MB: sorted %.4f s, created + written %.4f s (the largest, %.1f MB: %.4f s), closed %.4f s; opens: %.4f s summed, slowest %.4f s; open + write summed over the files %.4f s\n", nf, total / 1e6, tw1 - tw0, tw2 - tw1,
                nf ? o->sizes[order[0]] / 1e6 : 0.0, t_big, now() - tw2, t_open_sum, t_open_max, t_wr_sum);
    }
    return rc;
}

void crass_outputs_free(crass_outputs *o) { delete o; }

} // extern "C"
