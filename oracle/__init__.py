"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/cw_oracle.h).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
gym_craftingworld_amd/ (tests/test_host_logic.py: test_product_never_imports_oracle greps the product tree for such imports).
"""
from .oracle import OracleEnv, OracleBatch, build_oracle, TASK_LIST  # noqa: F401
