"""kart_amd -- the MI355X-native Kart hot path (FM-index seeding, chaining, NW gap closing) behind a C ABI (include/kart_amd.h, include/kart_host.h)."""
