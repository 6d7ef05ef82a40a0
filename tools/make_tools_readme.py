#!/usr/bin/env python3
"""Regenerate tools/README.md: one row per script with the first sentence(s) of its docstring / header comment."""
import ast, os
root = os.path.dirname(os.path.abspath(__file__))
rows = []
for f in sorted(os.listdir(root)):
    p = os.path.join(root, f)
    if f == "README.md" or os.path.isdir(p) or not f.endswith((".py", ".sh", ".hip")): continue
    txt = open(p, errors="replace").read()
    desc = ""
    if f.endswith(".py"):
        try: desc = " ".join((ast.get_docstring(ast.parse(txt)) or "").split())
        except Exception: pass
    if not desc:
        m = [l.lstrip("#/ ").strip() for l in txt.splitlines()[:6] if l.startswith(("#", "//")) and not l.startswith(("#!", "#include"))]
        desc = " ".join(m)
    rows.append((f, (desc[:260] + ("…" if len(desc) > 260 else "")).replace("|", "/")))
out = ["# tools/ — measurement and evidence scripts", "",
       "Nothing here is on the product path.  Every script below produced (or checks) a file under `profiles/`; `profiles/README.md` says which.", "",
       "Always-used: `check_isa.py` (post-build ISA lint, run by the CPU test tier), `profile_bench.sh` + `pmc_passes.sh` + `pmc_run.py` + `pmc_json.py` (the rocprofv3 passes behind",
       "`bench.py`'s `roofline` / `traffic`), `bench_configs.py` (`profiles/rNN_all_configs.jsonl`), `trace_case.sh` (per-kernel averages and gaps of a named case), `refresh_docs.py`",
       "(rewrites the README statements that `tests/test_host.py::test_docs_quote_the_recorded_numbers` checks), `make_tools_readme.py` (this index).  The rest are A/B and probe",
       "scripts, one per question; the first line of each says which.", "", "| script | what it measures |", "|---|---|"]
out += [f"| `{f}` | {d} |" for f, d in rows]
open(os.path.join(root, "README.md"), "w").write("\n".join(out) + "\n")
print(len(rows), "scripts")
