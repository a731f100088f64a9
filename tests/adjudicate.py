"""Near-tie adjudication for argmax-derived match rows (test infrastructure): the fp32 error model of the coarse stage
lives in oracle/error_model.py (also used by bench.py's in-run parity leg); this module re-exports it for the tests."""
from oracle.error_model import (U, ErrorModel, LocalErrorModel, _cells, _eps, assert_decidable_rows,  # noqa: F401
                                differing_rows_are_near_ties, differing_rows_are_near_ties_local)
