#!/usr/bin/env python3
"""The handles' option table (gpann_option_*; gp-ann_amd/csrc/shards.hip kKnobs) as a markdown table — INTEGRATION.md carries its output.
    python tools/gen_option_table.py            (needs the built library, no GPU)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-ann_amd"))
import pygpann  # noqa: E402

print("| option (name of the environment variable it is seeded from) | read | meaning |")
print("|---|---|---|")
for name, at_build, text in pygpann.options():
    print(f"| `{name}` | {'when the store / router is built' if at_build else 'by every call'} | {text.replace('|', chr(92) + '|')} |")
