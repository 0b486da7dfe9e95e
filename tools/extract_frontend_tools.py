#!/usr/bin/env python3
"""Extracts the frontend tool-call surface from the reference's page component into a constant table.

    python tools/extract_frontend_tools.py      (build container only: reads /root/reference)

Writes game_engine_amd/frontend_tools.json: {tool: [[param, type, required], ...]} for every
`useCopilotAction({ name, parameters: [...] })` of src/app/page.tsx (handlers :371-386, :892-2500;
e.g. createVotingPanel :1146-1157, markPlayerDead :1256-1262, clearCanvas :2418-2426) plus the
agent-side allow-list of frontend tool names (agent/game_agent_v2.py:144-192).  A table of names,
types and required flags - not the file's text."""
import json
import os
import re
import sys

REF = os.environ.get("GE_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "game_engine_amd", "frontend_tools.json")


def extract(page_text: str) -> dict:
    tools = {}
    for m in re.finditer(r"useCopilotAction\(\{", page_text):
        head = page_text[m.end(): m.end() + 6000]
        nm = re.search(r'name:\s*"(\w+)"', head)
        if not nm:
            continue
        handler_at = head.find("handler:")
        block = head[: handler_at if handler_at > 0 else len(head)]
        pm = re.search(r"parameters:\s*\[(.*)\]", block, re.S)
        params = []
        if pm:
            for p in re.finditer(r'\{\s*name:\s*"(\w+)"\s*,\s*type:\s*"([\w\[\]]+)"(?:\s*,\s*required:\s*(true|false))?', pm.group(1)):
                params.append([p.group(1), p.group(2), p.group(3) == "true"])
        tools[nm.group(1)] = params
    return tools


def allow_list(agent_text: str) -> list:
    m = re.search(r"FRONTEND_TOOL_ALLOWLIST\s*=\s*set\(\[(.*?)\]\)", agent_text, re.S)
    return sorted(set(re.findall(r'"(\w+)"', m.group(1)))) if m else []


def main():
    page = open(os.path.join(REF, "src", "app", "page.tsx"), encoding="utf-8").read()
    agent = open(os.path.join(REF, "agent", "game_agent_v2.py"), encoding="utf-8").read()
    table = {"source": "src/app/page.tsx useCopilotAction parameter lists; agent/game_agent_v2.py FRONTEND_TOOL_ALLOWLIST",
             "tools": extract(page), "allow_list": allow_list(agent)}
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(table, f, indent=1, sort_keys=True)
        f.write("\n")
    print(f"{len(table['tools'])} handlers, {len(table['allow_list'])} allow-listed names -> {OUT}")


if __name__ == "__main__":
    sys.exit(main())
