"""Stand-in for langchain_core.messages: plain containers."""


class BaseMessage:
    type = "base"

    def __init__(self, content="", **kw):
        self.content = content
        self.additional_kwargs = kw.pop("additional_kwargs", {})
        self.id = kw.pop("id", None)
        for k, v in kw.items():
            setattr(self, k, v)

    def __repr__(self):
        return f"{type(self).__name__}(content={self.content!r})"


class HumanMessage(BaseMessage):
    type = "human"


class SystemMessage(BaseMessage):
    type = "system"


class AIMessage(BaseMessage):
    type = "ai"

    def __init__(self, content="", tool_calls=None, **kw):
        super().__init__(content, **kw)
        self.tool_calls = list(tool_calls or [])


class ToolMessage(BaseMessage):
    type = "tool"

    def __init__(self, content="", tool_call_id=None, **kw):
        super().__init__(content, **kw)
        self.tool_call_id = tool_call_id
