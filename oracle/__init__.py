"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/ovo_oracle.h).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg. PARITY UNPINNED: the oracle is a
from-spec restatement; /root/reference holds no source to pin it against."""
