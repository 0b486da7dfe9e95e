"""Stand-in for langgraph.graph: records nodes; the harness walks them itself."""
END = "__end__"
START = "__start__"


class _Compiled:
    def __init__(self, g):
        self.nodes = dict(g.nodes)
        self.entry = g.entry


class StateGraph:
    def __init__(self, _state_type=None, **_kw):
        self.nodes = {}
        self.entry = None

    def add_node(self, name, fn):
        self.nodes[name] = fn

    def add_edge(self, *_a, **_kw):
        pass

    def set_entry_point(self, name):
        self.entry = name

    def compile(self, **_kw):
        return _Compiled(self)
