#!/usr/bin/env python3
"""One line per bench JSON file: ms/step, value, iterations, residual, phase times (and the accurate / cold legs)."""
import json
import sys
for path in sys.argv[1:]:
    try:
        d = json.load(open(path))
    except (OSError, ValueError) as e:
        print(path, "unreadable:", e)
        continue
    c = d["config"]
    acc = d.get("accurate")
    print("%s: %.2f ms/step = %.3g pts/s, %d it (coarse %d), true_rel %.2e, asm %.2f solve %.2f ms, roofline %.3f apply %.3f%s%s%s" % (
        path, d["ms_per_step"], d["value"], c["iterations"], c["coarse_iterations"], c["true_rel_residual"], c["assemble_ms"],
        c["solve_ms"], d["roofline"]["frac"], d.get("roofline_apply", {}).get("frac", 0.0),
        " err %.2e" % d["solution_rel_err"] if "solution_rel_err" in d else "",
        " cold %.1f ms" % d["cold_ms_per_step"] if "cold_ms_per_step" in d else "",
        " | accurate %.2f ms = %.3g pts/s, %d it, err %.2e" % (acc["ms_per_step"], acc["value"], acc["iterations"],
                                                              acc["solution_rel_err"]) if acc else ""))
