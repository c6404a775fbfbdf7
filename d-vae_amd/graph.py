"""A small directed-graph container with the python-igraph surface the D-VAE path touches.

The reference stores its DAGs as `igraph.Graph(directed=True)` objects with a
vertex attribute 'type' (`/root/reference/util.py:128-144, 255-275`,
`models.py:749-803`).  python-igraph is an optional dependency here: the models accept
any object with this surface (real igraph graphs included), and `decode()` / the loaders
return `DAG` objects, which answer the same calls:

    Graph(directed=True)  vcount  ecount  add_vertices  add_vertex(**attrs)  add_edge(s)
    predecessors  successors  neighbors(v, mode)  get_adjlist(mode)  get_edgelist
    get_adjacency().data  are_connected  is_dag  topological_sorting  copy
    indegree() / outdegree()      vs[i][attr]  vs[attr]  del vs[attr]  iteration (v.index,
    v.indegree(), v.outdegree(), v[attr])  vs.select(_outdegree_eq=k / _indegree_eq=k)

Vertices are dense ids 0..n-1; neighbour lists are kept as Python-int bitmasks (bit u of
`_pred[v]` <=> edge u -> v), which is also the packer's input format, and are reported in
ascending id order.  Multi-edges collapse (the D-VAE data never has them).
"""

IN = 2
OUT = 1
ALL = 3
_MODES = {"in": IN, "out": OUT, "all": ALL, IN: IN, OUT: OUT, ALL: ALL}


def _bits(mask):
    out = []
    i = 0
    while mask:
        if mask & 1:
            out.append(i)
        mask >>= 1
        i += 1
    return out


class Vertex(object):
    __slots__ = ("graph", "index")

    def __init__(self, graph, index):
        self.graph = graph
        self.index = index

    def __getitem__(self, name):
        return self.graph._attrs[name][self.index]

    def __setitem__(self, name, value):
        self.graph._column(name)[self.index] = value

    def attributes(self):
        return {k: col[self.index] for k, col in self.graph._attrs.items()}

    def indegree(self):
        return bin(self.graph._pred[self.index]).count("1")

    def outdegree(self):
        return bin(self.graph._succ[self.index]).count("1")

    def predecessors(self):
        return [Vertex(self.graph, u) for u in _bits(self.graph._pred[self.index])]

    def successors(self):
        return [Vertex(self.graph, u) for u in _bits(self.graph._succ[self.index])]


class VertexSeq(object):
    def __init__(self, graph, ids=None):
        self.graph = graph
        self._ids = ids

    def indices(self):
        return list(range(self.graph.vcount())) if self._ids is None else list(self._ids)

    def __len__(self):
        return self.graph.vcount() if self._ids is None else len(self._ids)

    def __iter__(self):
        return (Vertex(self.graph, i) for i in self.indices())

    def __getitem__(self, key):
        if isinstance(key, str):
            col = self.graph._attrs[key]
            return list(col) if self._ids is None else [col[i] for i in self._ids]
        if isinstance(key, slice):
            return VertexSeq(self.graph, self.indices()[key])
        ids = self.indices()
        return Vertex(self.graph, ids[key])

    def __setitem__(self, key, values):
        if not isinstance(key, str):
            raise TypeError("assign vertex attributes by name: g.vs['type'] = [...]")
        ids = self.indices()
        if not isinstance(values, (list, tuple)):
            values = [values] * len(ids)
        if len(values) != len(ids):
            raise ValueError("attribute list has %d entries for %d vertices" % (len(values), len(ids)))
        col = self.graph._column(key)
        for i, val in zip(ids, values):
            col[i] = val

    def __delitem__(self, key):
        del self.graph._attrs[key]

    def select(self, **conditions):
        ids = self.indices()
        g = self.graph
        for cond, val in conditions.items():
            if cond == "_outdegree_eq":
                ids = [i for i in ids if bin(g._succ[i]).count("1") == val]
            elif cond == "_indegree_eq":
                ids = [i for i in ids if bin(g._pred[i]).count("1") == val]
            elif cond.endswith("_eq") and not cond.startswith("_"):
                col = g._attrs[cond[:-3]]
                ids = [i for i in ids if col[i] == val]
            else:
                raise NotImplementedError("vs.select(%s=...)" % cond)
        return VertexSeq(g, ids)


class _Adjacency(object):
    def __init__(self, data):
        self.data = data

    def __iter__(self):
        return iter(self.data)

    def __getitem__(self, i):
        return self.data[i]


class DAG(object):
    def __init__(self, directed=True):
        if not directed:
            raise NotImplementedError("the D-VAE path only has directed graphs")
        self._pred = []
        self._succ = []
        self._attrs = {}

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_arrays(cls, types, pred_mask, n_vertices=None):
        """types[v], pred_mask[v] (bit u <=> edge u -> v) for the first n_vertices vertices."""
        n = len(types) if n_vertices is None else int(n_vertices)
        g = cls()
        g._pred = [int(pred_mask[v]) for v in range(n)]
        g._succ = [0] * n
        for v, m in enumerate(g._pred):
            for u in _bits(m):
                if u >= n:
                    raise ValueError("edge from vertex %d, graph has %d vertices" % (u, n))
                g._succ[u] |= 1 << v
        g._attrs["type"] = [int(types[v]) for v in range(n)]
        return g

    def _column(self, name):
        col = self._attrs.get(name)
        if col is None:
            col = [None] * len(self._pred)
            self._attrs[name] = col
        return col

    def add_vertices(self, n):
        self._pred.extend([0] * n)
        self._succ.extend([0] * n)
        for col in self._attrs.values():
            col.extend([None] * n)

    def add_vertex(self, name=None, **attrs):
        self.add_vertices(1)
        if name is not None:
            attrs["name"] = name
        for k, val in attrs.items():
            self._column(k)[-1] = val

    def add_edge(self, source, target):
        u, v = int(source), int(target)
        n = len(self._pred)
        if not (0 <= u < n and 0 <= v < n):
            raise ValueError("vertex id out of range")
        self._pred[v] |= 1 << u
        self._succ[u] |= 1 << v

    def add_edges(self, edges):
        for u, v in edges:
            self.add_edge(u, v)

    def copy(self):
        g = DAG()
        g._pred = list(self._pred)
        g._succ = list(self._succ)
        g._attrs = {k: list(col) for k, col in self._attrs.items()}
        return g

    # ------------------------------------------------------------------ queries
    @property
    def vs(self):
        return VertexSeq(self)

    def vcount(self):
        return len(self._pred)

    def ecount(self):
        return sum(bin(m).count("1") for m in self._pred)

    def is_directed(self):
        return True

    def predecessors(self, v):
        return _bits(self._pred[v])

    def successors(self, v):
        return _bits(self._succ[v])

    def neighbors(self, v, mode=ALL):
        mode = _MODES[mode]
        if mode == IN:
            return _bits(self._pred[v])
        if mode == OUT:
            return _bits(self._succ[v])
        return _bits(self._pred[v] | self._succ[v])

    def get_adjlist(self, mode=OUT):
        return [self.neighbors(v, mode) for v in range(len(self._pred))]

    def get_edgelist(self):
        return [(u, v) for u in range(len(self._succ)) for v in _bits(self._succ[u])]

    def get_adjacency(self):
        n = len(self._pred)
        return _Adjacency([[(self._succ[u] >> v) & 1 for v in range(n)] for u in range(n)])

    def are_connected(self, u, v):
        return bool((self._succ[int(u)] >> int(v)) & 1)

    def indegree(self):
        return [bin(m).count("1") for m in self._pred]

    def outdegree(self):
        return [bin(m).count("1") for m in self._succ]

    def topological_sorting(self):
        n = len(self._pred)
        indeg = self.indegree()
        order = [v for v in range(n) if indeg[v] == 0]
        i = 0
        while i < len(order):
            for w in _bits(self._succ[order[i]]):
                indeg[w] -= 1
                if indeg[w] == 0:
                    order.append(w)
            i += 1
        return order

    def is_dag(self):
        return len(self.topological_sorting()) == len(self._pred)

    def __repr__(self):
        return "DAG(n=%d, types=%s, edges=%s)" % (self.vcount(), self._attrs.get("type"), self.get_edgelist())


Graph = DAG  # `import dvae_amd.graph as igraph` reads like the reference's code
