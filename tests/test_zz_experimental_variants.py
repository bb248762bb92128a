"""-m gpu: the default-off kernel variants (CBIM_WINATTN_FWD2 / CBIM_WINATTN_BWD2) — correctness only, last file of the suite."""
import pytest

from tests import op_checks as oc

pytestmark = pytest.mark.gpu


def test_window_attention_two_queries_per_thread_variant(dev):
    """Experimental forward + backward (default off); correctness only — tools/run_round2_first.sh times it."""
    oc.check_window_attn_fwd2_variant(dev)
