"""Minimal duck-typed stand-in for python-igraph (TEST INFRASTRUCTURE ONLY).

python-igraph is not installed in this image and cannot be fetched.  The
reference model file (`/root/reference/models.py:3-9`) does `import igraph`
and touches only a handful of its methods (listed in SURVEY.md section 8b/8c):
Graph(directed=True), vcount, add_vertices, add_vertex(**attrs), add_edge,
predecessors, successors, get_adjlist(mode), copy, vs[i][attr] get/set,
vs[attr] get/set, del vs[attr], vs.select(_outdegree_eq=k), v.index.

This module is our own code.  It is placed on sys.path ONLY by
oracle/gen_golden.py (to import the reference and emit golden vectors) and by
tests / bench cpu_baseline that need a graph container.  The product package
never imports it; the product accepts any object with this duck-typed surface.

Neighbour lists are returned in ascending vertex-id order (real igraph returns
incidence order; only the floating-point summation order depends on it).
"""

IN = 2
OUT = 1
ALL = 3


class _Vertex(object):
    __slots__ = ("_g", "index")

    def __init__(self, g, index):
        self._g = g
        self.index = index

    def __getitem__(self, name):
        return self._g._attrs[name][self.index]

    def __setitem__(self, name, value):
        col = self._g._attrs.get(name)
        if col is None:
            col = [None] * self._g._n
            self._g._attrs[name] = col
        col[self.index] = value

    def indegree(self):
        return len(self._g._pred[self.index])

    def outdegree(self):
        return len(self._g._succ[self.index])


class _VertexSeq(object):
    def __init__(self, g, indices=None):
        self._g = g
        self._indices = indices

    def _ids(self):
        return range(self._g._n) if self._indices is None else self._indices

    def __len__(self):
        return len(self._ids())

    def __iter__(self):
        for i in self._ids():
            yield _Vertex(self._g, i)

    def __getitem__(self, key):
        if isinstance(key, str):
            col = self._g._attrs[key]
            return [col[i] for i in self._ids()]
        ids = self._ids()
        return _Vertex(self._g, ids[key])

    def __setitem__(self, key, values):
        if not isinstance(key, str):
            raise TypeError("vertex sequence assignment needs an attribute name")
        ids = list(self._ids())
        if not isinstance(values, (list, tuple)):
            values = [values] * len(ids)
        col = self._g._attrs.setdefault(key, [None] * self._g._n)
        for i, val in zip(ids, values):
            col[i] = val

    def __delitem__(self, key):
        del self._g._attrs[key]

    def select(self, **kw):
        ids = list(self._ids())
        for cond, val in kw.items():
            if cond == "_outdegree_eq":
                ids = [i for i in ids if len(self._g._succ[i]) == val]
            elif cond == "_indegree_eq":
                ids = [i for i in ids if len(self._g._pred[i]) == val]
            else:
                raise NotImplementedError(cond)
        return _VertexSeq(self._g, ids)


class Graph(object):
    def __init__(self, directed=True):
        if not directed:
            raise NotImplementedError("only directed graphs")
        self._n = 0
        self._pred = []
        self._succ = []
        self._attrs = {}

    # -- structure ---------------------------------------------------------
    def vcount(self):
        return self._n

    def ecount(self):
        return sum(len(p) for p in self._pred)

    def add_vertices(self, n):
        for _ in range(n):
            self._pred.append([])
            self._succ.append([])
        self._n += n
        for col in self._attrs.values():
            col.extend([None] * n)

    def add_vertex(self, **attrs):
        self.add_vertices(1)
        for k, val in attrs.items():
            col = self._attrs.setdefault(k, [None] * self._n)
            col[self._n - 1] = val

    def add_edge(self, u, v):
        u, v = int(u), int(v)
        if not (0 <= u < self._n and 0 <= v < self._n):
            raise ValueError("vertex id out of range")
        self._insert_sorted(self._pred[v], u)
        self._insert_sorted(self._succ[u], v)

    def add_edges(self, edges):
        for u, v in edges:
            self.add_edge(u, v)

    @staticmethod
    def _insert_sorted(lst, x):
        i = len(lst)
        while i > 0 and lst[i - 1] > x:
            i -= 1
        lst.insert(i, x)

    def predecessors(self, v):
        return list(self._pred[v])

    def successors(self, v):
        return list(self._succ[v])

    def get_adjlist(self, mode=OUT):
        if mode == IN:
            return [list(p) for p in self._pred]
        if mode == OUT:
            return [list(s) for s in self._succ]
        return [sorted(p + s) for p, s in zip(self._pred, self._succ)]

    def get_edgelist(self):
        return [(u, v) for u in range(self._n) for v in self._succ[u]]

    def copy(self):
        g = Graph(directed=True)
        g._n = self._n
        g._pred = [list(p) for p in self._pred]
        g._succ = [list(s) for s in self._succ]
        g._attrs = {k: list(col) for k, col in self._attrs.items()}
        return g

    @property
    def vs(self):
        return _VertexSeq(self)
