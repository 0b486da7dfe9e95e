"""Stand-in for langchain_core.runnables."""
RunnableConfig = dict
