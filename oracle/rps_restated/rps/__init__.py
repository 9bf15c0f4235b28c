"""ORACLE (test infrastructure, never shipped, never imported by marbler_amd/).

CPU restatement of the third-party `rps` package (robotarium_python_simulator,
the commit the reference pins only by prose: "6bb184e", /root/reference/README.md:11).
The package is NOT vendored in /root/reference and is not installed in this image,
so this is a restatement of its published algorithm from the spec recorded in
SURVEY.md Appendix A -- **parity against the real rps + cvxopt is unpinned**.

What IS pinned: the reference's own layers above rps (goal generation, the
sub-step driver, tracking, observations, rewards, termination) -- by running the
reference's Wrapper over this package (tests/golden/make_golden.py) and committing
the vectors under tests/golden/.

Only the symbols the reference's call sites use are provided:
  utilities/roboEnv.py:2,54,65,78,84-91,109-112,121
  utilities/controller.py:1-2,11-16,21-24
  utilities/misc.py:7,54
  scenarios/*/visualize.py:1
  scenarios/Warehouse/warehouse.py:93
"""
