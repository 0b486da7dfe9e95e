"""Stand-in for langchain.chat_models.

`init_chat_model(<vendor:model>)` never reaches a vendor: it returns the stub
model registered with `set_model_factory` (the harness's FixedPolicy).
"""
_factory = None


def set_model_factory(factory):
    global _factory
    _factory = factory


def init_chat_model(model_name, **_kw):
    if _factory is None:
        raise RuntimeError("no stub model registered (oracle.refharness.walker sets one)")
    return _factory(model_name)
