"""Stand-in for copilotkit: the agent state base is a plain dict."""
CopilotKitState = dict
