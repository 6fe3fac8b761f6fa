"""README.md / DESIGN.md quote measurements; the measurements live in profiles/.  tools/check_docs.py ties the two together
(the documents had drifted from the profiles -- a test count two rounds old, an update time from before its last
optimisation, rocprofv3's kernel average quoted as the driver-timed headline)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_headline_figures_of_readme_and_design_follow_from_the_committed_profiles():
    sys.path.insert(0, ROOT)
    from tools import check_docs
    problems = check_docs.check(verbose=False)
    assert not problems, "\n".join(problems)
