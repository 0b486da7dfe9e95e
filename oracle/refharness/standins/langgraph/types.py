"""Stand-in for langgraph.types.Command (subscriptable, carries goto/update)."""


class Command:
    def __init__(self, goto=None, update=None, **_kw):
        self.goto = goto
        self.update = update

    def __class_getitem__(cls, _item):
        return cls
