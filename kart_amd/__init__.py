"""kart_amd -- the MI355X-native Kart hot path (FM-index seeding, chaining, NW gap closing) behind a C ABI.

Importing the package settles one process-wide HIP setting BEFORE any HIP call can have been made through it: the runtime maps
its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With more streams than queues, two streams share a queue -- and on
this ROCm (7.2 / PyTorch 2.10+rocm7.0) a kernel was then seen to start before an asynchronous host-to-device copy enqueued in front
of it ON ITS OWN STREAM had landed: with eight stream lanes the first reads of a batch were seeded from the text of the batch
before (tools/stress_groups.py: 4-8 of 16 runs wrong with 6 or 8 lanes, 0 of 16 with GPU_MAX_HW_QUEUES=8, 0 with <= 5 lanes).
The product runs 8 lanes + 2 seeding groups = 10 streams, so it asks for 16 queues; the C library does the same in a constructor
for processes that load it directly (kart-amd), and kg_stream_open refuses more streams than the variable allows.  A value set by
the user is kept."""
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
