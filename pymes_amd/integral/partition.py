"""pymes/integral/partition.py:4-39 — the 16 named occ/virt blocks of V_pqrs.

Host arrays give numpy views (no copy, like the reference); on the device the same
16 blocks are packed into contiguous HBM arrays by ``Context.set_V_pqrs``."""

BLOCK_NAMES = ("abci", "iabj", "iajk", "aijk", "klij", "aibj", "ijak", "abic",
               "iajb", "abcd", "iabc", "aijb", "ijka", "aibc", "ijab", "abij")


def part_2_body_int(no, t_V_pqrs):
    rng = {True: slice(no, None), False: slice(0, no)}
    return {name: t_V_pqrs[tuple(rng[ch in "abcd"] for ch in name)] for name in BLOCK_NAMES}
