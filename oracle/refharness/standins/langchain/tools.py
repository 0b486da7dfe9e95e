"""Stand-in for langchain.tools: `@tool` keeps the function and exposes `.name`."""


class _Tool:
    def __init__(self, fn):
        self.fn = fn
        self.name = fn.__name__
        self.description = fn.__doc__ or ""
        self.__doc__ = fn.__doc__

    def __call__(self, *a, **kw):
        return self.fn(*a, **kw)

    def invoke(self, args):
        return self.fn(**args)


def tool(fn=None, **_kw):
    if fn is None:
        return lambda f: _Tool(f)
    return _Tool(fn)
