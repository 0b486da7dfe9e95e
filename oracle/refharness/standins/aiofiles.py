"""Stand-in for aiofiles.open: async context manager with async read()."""
import builtins


class _AFile:
    def __init__(self, path, mode, **kw):
        self._f = builtins.open(path, mode, **kw)

    async def __aenter__(self):
        return self

    async def __aexit__(self, *exc):
        self._f.close()
        return False

    async def read(self):
        return self._f.read()


def open(path, mode="r", **kw):
    return _AFile(path, mode, **kw)
