"""Host mirror of `zk_prover/src/merkle_sum_tree` (mst.rs:74-134, entry.rs:15-27,
utils/csv_parser.rs, utils/operation_helpers.rs:10-12) with the hashing on the GPU
(C ABI: sg_mst_leaves_dev / sg_mst_level_dev / sg_mst_build_dev).

Host logic kept here: CSV parsing, keccak256(username) -> field element, decimal balances ->
field elements, zero-entry padding to 2^depth, Merkle-proof extraction (tree.rs:85-137).
"""
from __future__ import annotations

import csv
import ctypes as C

import numpy as np

from . import ffi
from .utils import R_MODULUS

_KECCAK_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
              0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
              0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
              0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
              0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_KECCAK_ROT = [0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14]
_M64 = (1 << 64) - 1


def keccak256(data: bytes) -> bytes:
    """Keccak-256 as Ethereum uses it (`ethers::utils::keccak256`, entry.rs:21); lanes indexed x + 5y"""
    rate = 136
    msg = bytearray(data) + b"\x01"
    msg += bytes(-len(msg) % rate)
    msg[-1] |= 0x80
    st = [0] * 25
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            st[i] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        for rc in _KECCAK_RC:
            c = [st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20] for x in range(5)]
            for x in range(5):
                d = c[(x + 4) % 5] ^ (((c[(x + 1) % 5] << 1) | (c[(x + 1) % 5] >> 63)) & _M64)
                for y in range(0, 25, 5):
                    st[x + y] ^= d
            b = [0] * 25
            for x in range(5):
                for y in range(5):
                    v, r = st[x + 5 * y], _KECCAK_ROT[x + 5 * y]
                    b[y + 5 * ((2 * x + 3 * y) % 5)] = ((v << r) | (v >> (64 - r))) & _M64 if r else v
            for y in range(0, 25, 5):
                for x in range(5):
                    st[x + y] = b[x + y] ^ ((~b[(x + 1) % 5 + y]) & b[(x + 2) % 5 + y])
            st[0] ^= rc
    return b"".join(v.to_bytes(8, "little") for v in st[:4])


def _to_fr_bytes(v: int) -> bytes:
    return ((v % R_MODULUS) << 256).__mod__(R_MODULUS).to_bytes(32, "little")


def parse_csv_to_entries(path: str, n_currencies: int):
    """utils/csv_parser.rs: header `username,balance_<name>_<chain>,...`; one balance per
    currency column; raises if the column count differs from n_currencies (the reference's
    `try_into().unwrap()` panics there)."""
    with open(path, newline="") as f:
        first = f.readline()
    delim = ";" if ";" in first else ","
    with open(path, newline="") as f:
        rows = list(csv.reader(f, delimiter=delim))
    header, body = rows[0], rows[1:]
    if len(header) - 1 != n_currencies:
        raise ValueError(f"csv has {len(header) - 1} balance columns, N_CURRENCIES = {n_currencies}")
    cryptocurrencies = [tuple(h.split("_")[1:3]) for h in header[1:]]
    entries = [(r[0], [int(x) for x in r[1:1 + n_currencies]]) for r in body if r]
    return entries, cryptocurrencies


class MerkleSumTree:
    """MerkleSumTree<N_CURRENCIES, N_BYTES>: `from_csv`, `from_entries`, `root`, `generate_proof`."""

    def __init__(self, depth, n_currencies, entries, node_hashes, node_balances):
        self.depth, self.n_currencies, self.entries = depth, n_currencies, entries
        self._h, self._b = node_hashes, node_balances  # level-major numpy buffers

    @classmethod
    def from_csv(cls, path: str, n_currencies: int, n_bytes: int = 8):
        entries, _ = parse_csv_to_entries(path, n_currencies)
        return cls.from_entries(entries, n_currencies, n_bytes)

    @classmethod
    def from_entries(cls, entries, n_currencies: int, n_bytes: int = 8):
        import torch
        n = len(entries)
        if n == 0:
            raise ValueError("empty tree")
        for _, bal in entries:
            if any(b >= (1 << (8 * n_bytes)) for b in bal):
                raise ValueError("balance does not fit N_BYTES")  # range the circuit can prove (mst.rs)
        depth = max(0, (n - 1).bit_length())
        size = 1 << depth
        users = bytearray(32 * size)      # zero entries: username 0, balances 0 (entry.rs:30-38)
        bals = bytearray(32 * size * n_currencies)
        for i, (name, bal) in enumerate(entries):
            users[32 * i:32 * i + 32] = _to_fr_bytes(int.from_bytes(keccak256(name.encode()), "big"))
            for c, v in enumerate(bal):
                o = 32 * (i * n_currencies + c)
                bals[o:o + 32] = _to_fr_bytes(v)
        d_users = torch.from_numpy(np.frombuffer(bytes(users), dtype=np.uint8).copy()).cuda()
        d_bals = torch.from_numpy(np.frombuffer(bytes(bals), dtype=np.uint8).copy()).cuda()
        nodes = 2 * size - 1
        d_h = torch.empty(32 * nodes, dtype=torch.uint8, device="cuda")
        d_b = torch.empty(32 * nodes * n_currencies, dtype=torch.uint8, device="cuda")
        ffi.check(ffi.lib().sg_mst_build_dev(ffi.dev_ptr(d_users), ffi.dev_ptr(d_bals), C.c_uint32(depth),
                                             C.c_uint32(n_currencies), ffi.dev_ptr(d_h), ffi.dev_ptr(d_b),
                                             ffi.current_stream_ptr()))
        torch.cuda.synchronize()
        return cls(depth, n_currencies, list(entries), d_h.cpu().numpy(), d_b.cpu().numpy())

    def _level_offset(self, level: int) -> int:
        size = 1 << self.depth
        return sum(size >> l for l in range(level))

    def node(self, level: int, index: int):
        o = self._level_offset(level) + index
        nc = self.n_currencies
        return self._h[32 * o:32 * o + 32], self._b[32 * o * nc:32 * (o + 1) * nc]

    def root(self):
        return self.node(self.depth, 0)

    def generate_proof(self, index: int):
        """tree.rs:85-137: sibling (hash, balances) per level and the path bits"""
        sib, bits, idx = [], [], index
        for level in range(self.depth):
            bits.append(idx & 1)
            sib.append(self.node(level, idx ^ 1))
            idx >>= 1
        return {"leaf": self.node(0, index), "siblings": sib, "path_indices": bits, "root": self.root()}
