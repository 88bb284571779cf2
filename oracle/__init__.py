"""CPU checker for the TrueKNN hot path -- TEST INFRASTRUCTURE, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  ``owlraytracing_amd`` never does; its HIP path fails loudly when the extension is
missing instead of falling back to anything here.

PARITY UNPINNED: see the header of ``trueknn_oracle.c``.
"""
from .loader import (  # noqa: F401
    NEIGH_DTYPE,
    ORDER_ASCENDING,
    ORDER_DESCENDING,
    ORDER_SHUFFLED,
    OracleError,
    bruteforce_knn,
    build,
    dbscan,
    dbscan_auto,
    dbscan_auto_counts,
    dbscan_ball_check,
    dbscan_noise_count,
    dbscan_threaded,
    distance,
    num_threads,
    trueknn,
    trueknn_rows,
    trueknn_per_query,
)
