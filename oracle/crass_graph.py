"""oracle/crass_graph.py — TEST INFRASTRUCTURE, not product code.

CPU restatement (plain Python, small inputs: <= 10^4 reads per DR group) of what crass does with the hand-off AFTER
findConsensusDRs — SURVEY 8f rows f-4 and f-3:

  f-4  spacer graph     NodeManager::addReadHolder/splitReadHolder/addCrisprNodes   src/crass/NodeManager.cpp:120-443
                        cleanGraph / clearBubbles                                  :689-945   (+ CrisprNode.cpp)
                        buildSpacerGraph / cleanSpacerGraph / removeSpacerBubbles  :1063-1291 (+ SpacerInstance.cpp)
                        splitIntoContigs / walkFromCross                           :1293-1428
                        generateFlankers, getSpacerCountAndStats                   :2020-2068, :947-966
                        WorkHorse wrappers buildGraph ... removeLowConfidenceNodeManagers   WorkHorse.cpp:454-577,1642-1729
  f-3  outputs          WorkHorse::outputResults / addDataToDOM / addMetadataToDOM  WorkHorse.cpp:1900-2249
                        NodeManager::addSpacersToDOM ... printAssemblyToDOM        NodeManager.cpp:1503-1753
                        NodeManager::dumpReads + ReadHolder::print                 :1447-1500, ReadHolder.cpp:1243-1268
                        printSpacerGraph / printSpacerKey / Rainbow                :1789-2018, Rainbow.cpp
                        crispr::xml::writer (tag / attribute names base.cpp:72-121; root crassDefines.h:105-106)

PARITY UNPINNED.  None of these reference files can be compiled in this image (NodeManager.h pulls Xerces-C through
writer.h; everything pulls the autoconf config.h), the reference has no test or golden file for them, and two things make
the reference's own output depend on more than its input:
  * CrisprNode keeps its edges in std::map<CrisprNode*, bool> (CrisprNode.h:66) and cleanGraph uses a
    std::multimap<CrisprNode*, CrisprNode*> (NodeManager.cpp:698): iteration order = heap ADDRESS order.  This
    restatement uses creation order (= node id order, what a bump allocator would give) wherever the reference iterates
    such a container.  It matters only where a graph has forks / bubbles / inactive edges.
  * the XML is serialised by Xerces-C 3.1.1 (DOMLSSerializer, format-pretty-print; third party, not in the tree): the
    text layout below (declaration line, two-space indent, blank line around first-level elements, attributes in
    name order, `/>` for empty elements) is what that serialiser is known to produce, restated from its documented behaviour.
What is pinned: structural invariants, SURVEY 8c's per-group read counts on the reference inputs (tests/), well-formedness
and a round trip of the XML through Python's parser.  The product (crass_amd/csrc/adapter/crass_graph.cpp) is compared with
this file byte for byte.

Only tests/ may import this module.
"""
import math


def _i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def make_spacer_key(back, front):
    """makeSpacerKey (SpacerInstance.h:83-93): int arithmetic that overflows for tokens > 214, stored as unsigned int"""
    if back < front:
        return _i32(_i32(back * 10000000) + front) & 0xFFFFFFFF
    return _i32(_i32(front * 10000000) + back) & 0xFFFFFFFF


def make_key(i, j):
    """makeKey macro (NodeManager.h:88): (i*100000)+j in int"""
    return _i32(_i32(i * 100000) + j)


def c_round(x):
    """C round(): half away from zero; NaN stays NaN"""
    if x != x or math.isinf(x):
        return x
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


F, B, JF, JB = "F", "B", "JF", "JB"     # CN_EDGE_FORWARD, _BACKWARD, _JUMPING_F, _JUMPING_B
REVERSE, FORWARD = 0, 1                  # SI_EdgeDirection


class StringCheck:
    """StringCheck.cpp:46-81: addString ALWAYS hands out a new token (first is 2); the string -> token side keeps the latest"""

    def __init__(self):
        self.next = 1
        self.t2s, self.s2t = {}, {}

    def add(self, s):
        self.next += 1
        self.t2s[self.next] = s
        self.s2t[s] = self.next
        return self.next

    def token(self, s):
        return self.s2t.get(s, 0)

    def string(self, t):
        return self.t2s[t]


class Read:
    """the ReadHolder fields this stage reads (ReadHolder.h:440-451) + the spacer cutter (ReadHolder.cpp:813-952)"""

    def __init__(self, header, comment, seq, ss):
        self.header, self.comment, self.seq, self.ss = header, comment or b"", seq, list(ss)
        self.next = 0

    def first_spacer(self):
        self.next = 0
        return self.next_spacer()

    def next_spacer(self):
        """-> the spacer string, or None where the reference returns false"""
        ss, seq = self.ss, self.seq
        if self.next > len(ss) - 1:
            return None
        if self.next == 0:
            if ss[0] != 0:
                self.next = 1
                return seq[:ss[0]]
            start = ss[1] + 1
            if start > len(seq):
                raise IndexError("substring_exception")
            out = (seq[start:ss[2]] if ss[2] >= start else seq[start:]) if len(ss) > 2 else seq[start:]
            self.next = 3
            return out
        v = ss[self.next]
        if self.next == len(ss) - 1:
            if v < len(seq) - 1:
                self.next += 2
                return seq[v + 1:]
            return None
        start = v + 1
        if start > len(seq):
            raise IndexError("substring_exception")
        length = ss[self.next + 1] - start
        out = seq[start:start + length] if length >= 0 else seq[start:]     # (a negative length is a huge size_t)
        self.next += 2
        return out


class Node:
    """CrisprNode (CrisprNode.h:68-199, CrisprNode.cpp)"""

    def __init__(self, nid):
        self.id = nid
        self.attached, self.forward, self.coverage = True, True, 1
        self.edges = {F: {}, B: {}, JF: {}, JB: {}}          # partner id -> active; iterated in id (= creation) order
        self.rank = {F: 0, B: 0, JF: 0, JB: 0}
        self.headers = []

    def add_edge(self, other, t):
        if other.id not in self.edges[t]:
            self.edges[t][other.id] = True
            self.rank[t] += 1

    def total_rank(self):
        return self.rank[B] + self.rank[F] + self.rank[JF] + self.rank[JB]

    def inner_rank(self):
        return self.rank[B] + self.rank[F]

    def jumping_rank(self):
        return self.rank[JF] + self.rank[JB]


class Spacer:
    """SpacerInstance (SpacerInstance.h:95-174, SpacerInstance.cpp)"""

    def __init__(self, sid, leader, last):
        self.id, self.leader, self.last = sid, leader, last
        self.count, self.contig, self.attached, self.flanker = 1, 0, False, False
        self.edges = []                                       # [target Spacer, direction] in insertion order

    def rank(self):
        return len(self.edges)

    def find(self, other):
        return any(e[0] is other for e in self.edges)


class NodeManager:
    """one DR group's graph (NodeManager.h:124-299)"""

    KMER = 7                    # CRASS_DEF_NODE_KMER_SIZE (crassDefines.h:111)

    def __init__(self, dr, out):
        self.dr = dr
        self.sc = StringCheck()
        self.nodes = {}                                       # token -> Node (std::map: ascending token)
        self.spacers = {}                                     # SpacerKey -> Spacer (std::map: ascending unsigned key)
        self.reads = []
        self.next_contig = 0
        self.stats = []
        self.flankers = []
        self.out = out                                        # stdout lines the reference prints on the way
        self.rainbow = Rainbow()

    def _spacers(self):
        return [self.spacers[k] for k in sorted(self.spacers)]

    def _edge_nodes(self, node, t):
        return [self.nodes[i] for i in sorted(node.edges[t])]

    # ---- NodeManager.cpp:120-443 ----
    def add_read(self, rh):
        if self.split(rh):
            self.reads.append(rh)
            return True
        return False

    def split(self, rh):
        prev = [None]
        hst = self.sc.add(rh.header)
        ws = rh.first_spacer()
        if ws is None:
            return False
        if rh.ss[0] == 0:
            self.add_nodes(prev, ws, hst)
        else:
            self.add_second(prev, ws, hst)
        if len(rh.seq) == rh.ss[-1] + 1:
            while True:
                ws2 = rh.next_spacer()
                if ws2 is None:
                    break
                ws = ws2
                self.add_nodes(prev, ws, hst)
        else:
            while rh.next < len(rh.ss) - 1:
                ws2 = rh.next_spacer()
                if ws2 is not None:                           # (return value ignored: a stale string would be re-used)
                    ws = ws2
                self.add_nodes(prev, ws, hst)
            ws2 = rh.next_spacer()
            if ws2 is not None:
                self.add_first(prev, ws2, hst)
        return True

    def _node(self, kmer, forward):
        st = self.sc.token(kmer)
        if st == 0:
            st = self.sc.add(kmer)
            n = Node(st)
            if not forward:
                n.forward = False
            self.nodes[st] = n
        else:
            n = self.nodes[st]
            n.coverage += 1
        return n

    def add_nodes(self, prev, ws, hst):
        k = self.KMER
        if len(ws) < k:
            return
        n1 = self._node(ws[:k], True)
        n2 = self._node(ws[len(ws) - k:], False)
        n1.headers.append(hst)
        n2.headers.append(hst)
        if prev[0] is not None:
            if make_spacer_key(n1.id, prev[0].id) not in self.spacers:
                prev[0].add_edge(n1, JF)
                n1.add_edge(prev[0], JB)
        key = make_spacer_key(n1.id, n2.id)
        if key not in self.spacers:
            st = self.sc.token(ws)
            if st == 0:
                st = self.sc.add(ws)
            self.spacers[key] = Spacer(st, n1, n2)
            n1.add_edge(n2, F)
            n2.add_edge(n1, B)
        else:
            self.spacers[key].count += 1
        prev[0] = n2

    def add_second(self, prev, ws, hst):
        k = self.KMER
        if len(ws) < k:
            return
        n2 = self._node(ws[len(ws) - k:], False)
        n2.headers.append(hst)
        prev[0] = n2

    def add_first(self, prev, ws, hst):
        k = self.KMER
        if len(ws) < k:
            return
        n1 = self._node(ws[:k], True)
        n1.headers.append(hst)
        if prev[0] is not None:
            if make_spacer_key(n1.id, prev[0].id) not in self.spacers:
                prev[0].add_edge(n1, JF)
                n1.add_edge(prev[0], JB)

    # ---- CrisprNode.cpp:181-214: detach with the reference's same-type bookkeeping ----
    def set_attach(self, node, state):
        for t in (F, B, JF, JB):
            for pid in sorted(node.edges[t]):
                p = self.nodes[pid]
                if (node.edges[t][pid] != state) and p.attached:
                    p.edges[t][node.id] = state               # the partner's list of the SAME type (CrisprNode.cpp:189-190)
                    node.edges[t][pid] = state
                    p.rank[t] += 1 if state else -1
                    if p.total_rank() == 0:
                        p.attached = False
        node.attached = state

    def discounted_coverage(self, node):
        """CrisprNode::getDiscountedCoverage (CrisprNode.cpp:131-178)"""
        cm = {h: 0 for h in node.headers}
        for t in ((F, JB) if node.forward else (JF, B)):
            for pid in sorted(node.edges[t]):
                if not node.edges[t][pid]:
                    continue
                for h in self.nodes[pid].headers:
                    if h in cm:
                        cm[h] += 1
        return sum(1 for v in cm.values() if v > 1)

    def find_all(self):
        caps, other = [], []
        for t in sorted(self.nodes):
            n = self.nodes[t]
            if n.attached:
                (caps if n.total_rank() == 1 else other).append(n)
        return caps, other

    def find_caps_at(self, forward, inner, strict, q):
        caps = []
        if q.attached:
            t = (F if inner else JF) if forward else (B if inner else JB)
            for pid in sorted(q.edges[t]):
                if q.edges[t][pid]:
                    a = self.nodes[pid]
                    if a.total_rank() == 1:
                        caps.append(a)
                    elif strict:
                        return []
        return caps

    # ---- NodeManager::cleanGraph (NodeManager.cpp:689-858) ----
    def clean_graph(self):
        some = True
        while some:
            some = False
            fork = []                                          # multimap joining node -> cap
            detach = []
            caps, _ = self.find_all()
            for n in caps:
                if n.inner_rank() == 0:
                    el = JF if n.rank[JF] != 0 else JB
                    first = self._edge_nodes(n, el)[0]
                    if first.total_rank() != 2:
                        detach.append(n)
                else:
                    if n.rank[F] != 0:
                        el, is_forward = F, False
                    else:
                        el, is_forward = B, True
                    j = self._edge_nodes(n, el)[0]
                    if j.total_rank() != 2:
                        if len(self.find_caps_at(is_forward, True, True, j)) > 1:
                            fork.append((j, n))
                        else:
                            detach.append(n)
            fork.sort(key=lambda jn: jn[0].id)                # (stable: equal keys keep their insertion order)
            best = {}
            for j, n in fork:
                if j.id not in best or best[j.id].coverage < n.coverage:
                    best[j.id] = n
            for j, n in fork:
                if best[j.id] is not n:
                    detach.append(n)
            if detach:
                some = True
            for n in detach:
                self.set_attach(n, False)
            _, other = self.find_all()
            for n in other:
                r = n.total_rank()
                if r == 2:
                    if not (n.inner_rank() and n.jumping_rank()):
                        self.set_attach(n, False)
                        some = True
                elif r in (0, 1):
                    pass
                else:
                    if n.inner_rank() != 1:
                        if self.clear_bubbles(n, F):
                            some = True
                    if n.jumping_rank() != 1:
                        if self.clear_bubbles(n, JF):
                            some = True
        return 0

    @staticmethod
    def _next_key(d, cur):
        """the next key of a std::map during iteration: the smallest key greater than the current one — entries that
        CrisprNode::setAttach inserts behind the iterator while the loop runs are visited, those before it are not"""
        ks = [k for k in d if k > cur]
        return min(ks) if ks else None

    def clear_bubbles(self, root, t):
        some = False
        opposite = {B: JB, F: JF, JB: B, JF: F}[t]
        bubble = {}
        ek = self._next_key(root.edges[t], -1)
        while ek is not None:
            e = self.nodes[ek]
            if e.attached:
                k2 = self._next_key(e.edges[opposite], -1)
                while k2 is not None:
                    e2 = self.nodes[k2]
                    if e2.attached:
                        key = make_key(root.id, e2.id)
                        if key not in bubble:
                            bubble[key] = e.id
                        else:
                            first = self.nodes[bubble[key]]
                            if self.discounted_coverage(first) > self.discounted_coverage(e):
                                self.set_attach(e, False)
                                some = True
                            else:
                                self.set_attach(first, False)
                                some = True
                                bubble[key] = e.id
                    k2 = self._next_key(e.edges[opposite], k2)
            ek = self._next_key(root.edges[t], ek)
        return some

    # ---- spacer graph (NodeManager.cpp:1063-1291) ----
    def spacer_attached(self, s):
        """SpacerInstance::isAttached (SpacerInstance.h:121-128) incl. its message"""
        if s.leader.attached and (s.last.attached ^ s.attached):
            self.out.append("Spacer %d has asynchronous attached state" % s.id)
        return s.attached

    def build_spacer_graph(self):
        for s in self._spacers():
            if s.last.attached and s.leader.attached:
                s.attached = True
                for q in self._edge_nodes(s.last, JF):
                    if q.attached and q.forward:
                        for e in self._edge_nodes(q, F):
                            if e.attached:
                                nxt = self.spacers[make_spacer_key(e.id, q.id)]
                                if nxt is not s:
                                    s.edges.append([nxt, FORWARD])
                                    nxt.edges.append([s, REVERSE])
            else:
                s.attached = False
        return 0

    def detach_spacer(self, s):
        """SpacerInstance::detachFromSpacerGraph (SpacerInstance.cpp:160-188); SI_Attached stays as it is"""
        if s.rank() == 0:
            return
        for tgt, _ in s.edges:
            for i, e in enumerate(tgt.edges):
                if e[0] is s:
                    del tgt.edges[i]
                    break
            else:
                raise RuntimeError("detachSpecificSpacer: no return edge (the reference would leave dangling edges)")
        s.edges = []

    @staticmethod
    def is_fur(s):
        return s.rank() == 1 and any(t.rank() > 2 for t, _ in s.edges)

    @staticmethod
    def is_viable(s):
        if s.rank() < 2:
            return True
        return any(d == REVERSE for _, d in s.edges) and any(d != REVERSE for _, d in s.edges)

    def clean_spacer_graph(self):
        cleaned = True
        while cleaned:
            cleaned = False
            for s in self._spacers():
                if self.spacer_attached(s) and self.is_fur(s):
                    self.detach_spacer(s)
                    cleaned = True
            for s in self._spacers():
                if self.spacer_attached(s) and not self.is_viable(s):
                    self.detach_spacer(s)
                    cleaned = True
            self.remove_spacer_bubbles()
        return 0

    def remove_spacer_bubbles(self):
        bubble = {}
        detach = []
        for cur in self._spacers():
            if not self.spacer_attached(cur):
                continue
            if cur.rank() < 2:
                continue
            r_sp = [t for t, d in cur.edges if d == REVERSE]
            f_sp = [t for t, d in cur.edges if d != REVERSE]
            for rs in r_sp:
                for fs in f_sp:
                    key = make_spacer_key(rs.id, fs.id)
                    if key not in bubble:
                        bubble[key] = cur
                    else:
                        if rs.find(cur) and rs.find(bubble[key]):
                            continue
                        self.out.append("Coverage test: %d : %d" % (bubble[key].id, cur.id))
                        if bubble[key].count < cur.count:
                            detach.append(bubble[key])
                            bubble[key] = cur
                        elif cur.count < bubble[key].count:
                            detach.append(cur)
                        elif bubble[key].rank() < cur.rank():
                            detach.append(bubble[key])
                            bubble[key] = cur
                        else:
                            detach.append(cur)
        for s in detach:
            self.detach_spacer(s)

    # ---- contigs (NodeManager.cpp:573-686, 1293-1428) ----
    def split_into_contigs(self):
        walk = [None, None, None]                             # WalkingManager: first, second, wanted edge type
        start = [s for s in self._spacers() if self.spacer_attached(s) and s.rank() == 1]
        cross = []
        for cap in start:
            cur = []
            self.next_contig += 1
            if self.edge_from_cap(walk, cap):
                prev = [None]
                while True:
                    if prev[0] is not None:
                        cur.append(prev[0])
                    if not self.step(walk, prev):
                        break
                cur.append(walk[0])
                if walk[1].rank() == 1:
                    cur.append(walk[1])
                else:
                    cross.append(walk[1])
                for s in cur:
                    s.contig = self.next_contig
        self.next_contig += 1
        self.walk_from_cross(cross)
        return 0

    def edge_from_cap(self, walk, cur):
        if cur.rank() == 1:
            for tgt, d in cur.edges:
                if self.spacer_attached(tgt):
                    if tgt.contig == 0:
                        walk[1], walk[0], walk[2] = tgt, cur, d
                    else:
                        cur.contig = tgt.contig
                        return False
                else:
                    return False
        else:
            return False
        return not (walk[0] is None or walk[1] is None)

    def edge_from_cross(self, walk, cur):
        if cur.rank() == 2:
            for tgt, d in cur.edges:
                if self.spacer_attached(tgt):
                    if tgt.contig == 0:
                        walk[1], walk[0], walk[2] = tgt, cur, d
                        return True
                else:
                    return False
        else:
            return False
        return not (walk[0] is None or walk[1] is None)

    def step(self, walk, prev):
        if walk[1].rank() == 2:
            for tgt, d in walk[1].edges:
                if self.spacer_attached(tgt) and d == walk[2] and tgt.id != walk[0].id and tgt.contig == 0:
                    prev[0] = walk[0]
                    walk[0], walk[1] = walk[1], tgt
                    return True
        return False

    def walk_from_cross(self, cross):
        walk = [None, None, None]                             # (a fresh WalkingManager; its members start indeterminate in the reference)
        i = 0
        while i < len(cross):
            c = cross[i]
            c.contig = self.next_contig
            self.next_contig += 1
            for tgt, _ in list(c.edges):
                if self.spacer_attached(tgt) and tgt.contig == 0:
                    if self.edge_from_cross(walk, tgt):
                        cur = []
                        prev = [None]
                        while True:
                            if prev[0] is not None:
                                cur.append(prev[0])
                            if not self.step(walk, prev):
                                break
                        if walk[1].rank() == 1 and self.spacer_attached(walk[1]):
                            cur.append(walk[1])
                        elif walk[1].contig == 0 and self.spacer_attached(walk[1]):
                            cur.append(walk[0])
                            cross.append(walk[1])
                        for s in cur:
                            s.contig = self.next_contig
                        self.next_contig += 1
                    else:
                        cross.append(tgt)
            i += 1

    # ---- stats / flankers (NodeManager.cpp:947-966, 2020-2068; StatsManager.h) ----
    def spacer_count_and_stats(self, show_detached=False, exclude_flankers=True):
        n = 0
        for s in self._spacers():
            if show_detached or self.spacer_attached(s):
                if exclude_flankers and s.flanker:
                    continue
                self.stats.append(len(self.sc.string(s.id)))
                n += 1
        return n

    def stats_mean(self):
        return sum(self.stats) // len(self.stats)              # size_t arithmetic

    def stats_stdev(self):
        avg = float(self.stats_mean())
        return math.sqrt(sum((float(v) - avg) * (float(v) - avg) for v in self.stats) / len(self.stats))

    def generate_flankers(self, show_detached=False):
        n = self.spacer_count_and_stats()
        if n >= 3:
            stdev = self.stats_stdev()
            mean = int(self.stats_mean())
            lower = int(mean - (stdev * 1.5))                  # double -> int: towards zero
            upper = int(mean + (stdev * 1.5))
            if stdev > 1:
                for s in self._spacers():
                    if show_detached or (s.leader.attached and s.last.attached):
                        ln = len(self.sc.string(s.id))
                        if ln > upper or ln < lower:
                            s.flanker = True
                            self.flankers.append(s)
        self.stats = []

    # ---- outputs ----
    def spacer_label(self, s, long_desc=False):
        pre = "fl_" if s.flanker else "sp_"
        if long_desc:
            lab = "%s%d_%s_%d" % (pre, s.id, self.sc.string(s.id).decode("latin-1"), s.count)
        else:
            lab = "%s%d_%d" % (pre, s.id, s.count)
        return lab + "_C%d" % s.contig

    def set_spacer_colour_limits(self):
        mx, mn = 0.0, 10000000.0
        for s in self._spacers():
            c = s.count
            if c > mx:
                mx = float(c)
            elif c < mn:
                mn = float(c)
        self.rainbow.set_type("BLUE_RED")
        self.rainbow.set_limits(mn, mx, int(mx - mn) + 1)      # coverageBins == -1 (crassDefines.h:135)

    def spacer_graph_text(self, title, long_desc=False, show_singles=False):
        """NodeManager::printSpacerGraph (NodeManager.cpp:1883-1954): None where it returns false"""
        self.set_spacer_colour_limits()
        head = "digraph %s {\n" % title
        any_spacer = False
        sel = []
        for s in self._spacers():
            if self.spacer_attached(s) and (show_singles or s.rank() != 0):
                any_spacer = True
                sel.append(s)
                lab = self.spacer_label(s, long_desc)
                col = self.rainbow.colour(s.count)
                head += "\t\t%s [ color = \"#%s\", fillcolor=\"#%s\", style= filled, shape=%s];\n" % (
                    lab, col, col, "diamond" if s.flanker else "circle")
        if not any_spacer:
            return None
        for s in sel:
            lab = self.spacer_label(s, long_desc)
            for tgt, d in s.edges:
                if self.spacer_attached(tgt) and d == FORWARD and (show_singles or tgt.rank() != 0):
                    head += "\t\t%s -> %s [ len=2 ];\n" % (lab, self.spacer_label(tgt, long_desc))
        return head + "\n}\n"

    def spacer_key_text(self, cluster_number, group_name, steps=10):
        """NodeManager::printSpacerKey (NodeManager.cpp:1996-2018)"""
        t = ("\tsubgraph cluster_%d\t{\n\t\t\"%s\" [ fillcolor = \"white\" shape = \"record\" label =<<table border=\"0\" cellborder=\"0\" "
             "cellpadding=\"0\" bgcolor=\"white\"><tr><td>%s</td></tr>" % (cluster_number, group_name, group_name))
        ul, ll = self.rainbow.ub, self.rainbow.lb
        step = (ul - ll) / (steps - 1)
        if step < 1:
            step = 1
        i = ll
        while i <= ul:
            this = int(i)
            t += "<tr><td bgcolor=\"#%s\" align=\"center\" colspan=\"2\"><font color=\"white\">%d</font></td></tr>" % (self.rainbow.colour(this), this)
            i += step
        return t + "</table>> ];\n\t}\n"

    def dump_reads_text(self, show_detached=True):
        """NodeManager::dumpReads (NodeManager.cpp:1447-1500) + ReadHolder::print (ReadHolder.cpp:1243-1268)"""
        names = set()
        for s in self._spacers():
            if show_detached or (s.leader.attached and s.last.attached):
                for n in (s.leader, s.last):
                    for h in n.headers:
                        names.add(self.sc.string(h))
        out = b""
        for r in self.reads:
            if r.header in names:
                out += b">" + r.header + ((b" " + r.comment) if len(r.comment) > 0 else b"") + b"\n" + r.seq + b"\n"
        return out


class Rainbow:
    """Rainbow.cpp:47-208 (heat-map colours of the .gv files)"""
    PI, DIV = 3.1415927, 0.6666666666

    def __init__(self):
        self.lb = self.ub = 0.0
        self.res = 0
        self.upper_scale = self.lower_scale = 0.0
        self.mult = self.tick = 0.0
        self.set_type("BLUE_RED")
        self.set_limits(0.0, 1.0, 10)

    def set_type(self, t):
        assert t == "BLUE_RED"                               # CRASS_DEF_GRAPH_COLOUR (crassDefines.h:136)
        self.red_off, self.green_off, self.blue_off = self.DIV * self.PI, self.DIV * self.PI * 2, 0.0
        self.lower_scale, self.upper_scale = 0.0, self.DIV * self.PI
        self.mult = self._div(self.upper_scale - self.lower_scale, self.ub - self.lb)

    @staticmethod
    def _div(a, b):
        if b == 0:
            return float("nan") if a == 0 else math.copysign(float("inf"), a) * (1 if math.copysign(1, b) > 0 else -1)
        return a / b

    def set_limits(self, lb, ub, res):
        self.lb, self.ub, self.res = float(lb), float(ub), res
        self.mult = self._div(self.upper_scale - self.lower_scale, self.ub - self.lb)
        self.tick = self._div(self.ub - self.lb, float(self.res - 1))

    def _rgb(self, v):
        if v != v or math.isinf(v):
            return "00"                                       # (int)(NaN) is INT_MIN on x86-64: "0 >= rgb"
        rgb = int(v)
        if rgb <= 0:
            return "00"
        return ("0%x" % rgb) if rgb < 16 else ("%x" % rgb)

    def colour(self, value):
        value = float(value)
        if self.res == -1:
            return "000000"
        if value > self.ub or value < self.lb:
            return "000000"
        norm = c_round(self._div(value, self.tick)) * self.tick
        scaled = (norm - self.lb) * self.mult + self.lower_scale

        def val(x):
            return (math.cos(x) + 0.5) * self.DIV if x == x and not math.isinf(x) else float("nan")
        return self._rgb(c_round(val(scaled - self.red_off) * 255)) + "00" + self._rgb(c_round(val(scaled - self.blue_off) * 255))


# ---- XML text as Xerces-C's DOMLSSerializer (format-pretty-print) lays it out ----
def _esc_attr(b):
    s = b.decode("latin-1") if isinstance(b, (bytes, bytearray)) else str(b)
    out = []
    for ch in s:
        o = ord(ch)
        if ch == "&":
            out.append("&amp;")
        elif ch == "<":
            out.append("&lt;")
        elif ch == '"':
            out.append("&quot;")
        elif o in (9, 10, 13):
            out.append("&#x%X;" % o)
        else:
            out.append(ch)
    return "".join(out)


def _esc_text(b):
    s = b.decode("latin-1") if isinstance(b, (bytes, bytearray)) else str(b)
    return s.replace("&", "&amp;").replace("<", "&lt;").replace(">", "&gt;")


class Elem:
    def __init__(self, tag, attrs=None, text=None):
        self.tag, self.attrs, self.text, self.kids = tag, dict(attrs or {}), text, []

    def add(self, e):
        self.kids.append(e)
        return e

    def write(self, out, level):
        if level == 1:
            out.append("\n")                                  # format-pretty-print-1st-level: a blank line before the element
        out.append("\n" + "  " * level + "<" + self.tag)
        for k in sorted(self.attrs):                          # DOMAttrMapImpl keeps attributes sorted by name
            out.append(' %s="%s"' % (k, _esc_attr(self.attrs[k])))
        if self.text is not None:
            out.append(">" + _esc_text(self.text) + "</" + self.tag + ">")
        elif self.kids:
            out.append(">")
            for k in self.kids:
                k.write(out, level + 1)
            if level == 0:
                out.append("\n")
            out.append("\n" + "  " * level + "</" + self.tag + ">")
        else:
            out.append("/>")


def xml_text(root):
    out = ['<?xml version="1.0" encoding="ISO8859-1" standalone="no" ?>']
    root.write(out, 0)
    out.append("\n")
    return "".join(out).encode("latin-1")


def run(groups, outdir=b"./", timestamp="01_01_2026_000000", cmdline="crass ", cwd="/cwd", log_to_screen=True, cov_cutoff=3,
        package="crass", version="1.0.1"):
    """WorkHorse::doWork after parseSeqFiles (WorkHorse.cpp:196-316).
    groups: [(gid, true_dr, [Read, ...])] in ascending GID; the reads in buildGraph's order (WorkHorse.cpp:454-505: for every
    token of the group's cluster, mReads[token] in list order).
    -> dict: files {name: bytes}, stdout [str], kept [gid] (groups in the XML), spacers {gid: attached non-flanker count}"""
    outdir_s = outdir.decode() if isinstance(outdir, bytes) else outdir
    out = []
    nms = {}
    # buildGraph: mDRs is keyed by the true DR string: a later group with the same DR replaces the earlier NodeManager
    by_dr = {}
    for gid, dr, reads in groups:
        nm = NodeManager(dr, out)
        by_dr[dr] = nm
        for r in reads:
            nm.add_read(r)
    for gid, dr, reads in groups:
        nms[gid] = by_dr[dr]
    order = [g for g, _, _ in groups]
    dr_of = {g: d for g, d, _ in groups}
    alive = dict(nms)

    def each(fn):
        for g in order:
            if alive.get(g) is not None:
                fn(alive[g])
    each(lambda nm: nm.clean_graph())
    for dr in sorted(by_dr):                                 # makeSpacerGraphs .. splitIntoContigs walk mDRs (string order)
        by_dr[dr].build_spacer_graph()
    for dr in sorted(by_dr):
        by_dr[dr].clean_spacer_graph()
    for dr in sorted(by_dr):
        by_dr[dr].split_into_contigs()
    each(lambda nm: nm.generate_flankers())
    # removeLowConfidenceNodeManagers (WorkHorse.cpp:544-573)
    for g in order:
        nm = alive.get(g)
        if nm is None:
            continue
        if nm.spacer_count_and_stats(False) < cov_cutoff:
            alive[g] = None
        elif nm.stats_stdev() > 6.0:
            alive[g] = None
    # outputResults (WorkHorse.cpp:1900-2038)
    files = {}
    name_prefix = outdir_s + package + ".crispr"
    keys = "digraph Keys {\n"
    root = Elem("crispr", {"version": "1.1"})
    kept = []
    cluster_number = 0
    for g in order:
        nm = alive.get(g)
        if nm is None:
            continue
        dr = dr_of[g].decode()
        gv = nm.spacer_graph_text(dr)
        if gv is None:
            alive[g] = None
            continue
        gv_name = "Spacers_%d_%s_spacers.gv" % (g, dr)
        fa_name = "Group_%d_%s.fa" % (g, dr)
        files[gv_name] = gv.encode("latin-1")
        keys += nm.spacer_key_text(cluster_number, name_prefix + str(g))
        cluster_number += 1
        files[fa_name] = nm.dump_reads_text(True)
        kept.append(g)
        grp = root.add(Elem("group", {"gid": "G%d" % g, "drseq": dr}))
        # <data> (WorkHorse.cpp:2040-2088)
        data = grp.add(Elem("data"))
        sources = data.add(Elem("sources"))
        drs = data.add(Elem("drs"))
        spacers = data.add(Elem("spacers"))
        flankers = data.add(Elem("flankers")) if nm.flankers else None
        all_sources = set()
        drs.add(Elem("dr", {"drid": "DR1", "seq": dr}))

        def src_tokens(s):
            return sorted(set(s.leader.headers) | set(s.last.headers))
        for s in nm._spacers():
            if (s.leader.attached and s.last.attached) and not s.flanker:
                toks = src_tokens(s)
                e = spacers.add(Elem("spacer", {"seq": nm.sc.string(s.id), "spid": "SP%d" % s.id, "cov": str(s.count)}))
                for t in toks:
                    e.add(Elem("source", {"soid": "SO%d" % t}))
                all_sources.update(toks)
        if flankers is not None:
            for s in nm.flankers:
                if s.leader.attached and s.last.attached:
                    toks = src_tokens(s)
                    e = flankers.add(Elem("flanker", {"seq": nm.sc.string(s.id), "flid": "FL%d" % s.id}))
                    for t in toks:
                        e.add(Elem("source", {"soid": "SO%d" % t}))
                    all_sources.update(toks)
        for t in sorted(all_sources):
            sources.add(Elem("source", {"accession": nm.sc.string(t), "soid": "SO%d" % t}))
        # <metadata> (WorkHorse.cpp:2090-2249)
        meta = grp.add(Elem("metadata"))
        prog = meta.add(Elem("program"))
        prog.add(Elem("name", text=package))
        prog.add(Elem("version", text=version))
        prog.add(Elem("command", text=cmdline))
        meta.add(Elem("notes", text="Run on " + timestamp))
        absdir = cwd + "/"
        if not log_to_screen:
            meta.add(Elem("file", {"type": "log", "url": absdir + outdir_s + package + "." + timestamp + ".log"}))
        meta.add(Elem("file", {"type": "data", "url": absdir + outdir_s + gv_name}))
        meta.add(Elem("file", {"type": "sequence", "url": absdir + outdir_s + fa_name}))
        # <assembly> (NodeManager.cpp:1560-1706)
        asm = grp.add(Elem("assembly"))
        for cn in range(1, nm.next_contig + 1):
            ce = asm.add(Elem("contig", {"cid": "C%d" % cn}))
            for s in nm._spacers():
                if s.contig == cn and nm.spacer_attached(s):
                    cs = ce.add(Elem("cspacer", {"spid": ("FL%d" if s.flanker else "SP%d") % s.id}))
                    fsp = bsp = ffl = bfl = None
                    for tgt, d in s.edges:
                        if nm.spacer_attached(tgt):
                            eid = ("FL%d" if s.flanker else "SP%d") % tgt.id
                            if d == FORWARD:
                                if tgt.flanker:
                                    ffl = ffl or Elem("fflankers")
                                    ffl.add(Elem("ff", {"flid": eid, "drconf": "0", "directjoin": "0"}))
                                else:
                                    fsp = fsp or Elem("fspacers")
                                    fsp.add(Elem("fs", {"drid": "DR1", "drconf": "0", "spid": eid}))
                            else:
                                if tgt.flanker:
                                    bfl = bfl or Elem("bflankers")
                                    bfl.add(Elem("bf", {"flid": eid, "drconf": "0", "directjoin": "0"}))
                                else:
                                    bsp = bsp or Elem("bspacers")
                                    bsp.add(Elem("bs", {"drid": "DR1", "drconf": "0", "spid": eid}))
                    for e in (bsp, fsp, bfl, ffl):
                        if e is not None:
                            cs.add(e)
    out.append("[%s_graphBuilder]: %d CRISPRs found!" % (package, len(kept)))
    files[package + ".crispr"] = xml_text(root)
    files["%s.%s.keys.gv" % (package, timestamp)] = (keys + "\n}\n").encode("latin-1")
    spacers_n = {g: sum(1 for s in alive[g]._spacers() if s.attached and not s.flanker) for g in kept}
    return dict(files=files, stdout=out, kept=kept, spacers=spacers_n, managers={g: alive[g] for g in kept})
