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
    """Keccak-256 as Ethereum uses it (`ethers::utils::keccak256`, entry.rs:21): the library's host routine
    (sg_keccak256; the EVM transcript hashes a few KiB per proof)"""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    out = np.zeros(32, dtype=np.uint8)
    ffi.check(ffi.lib().sg_keccak256(ffi.ptr(buf) if buf.size else None, C.c_size_t(buf.size), ffi.ptr(out)))
    return out.tobytes()


def keccak256_python(data: bytes) -> bytes:
    """the same with Python integers (lanes indexed x + 5y): the independent twin the library routine is checked against"""
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


def big_intify_username(username: str) -> int:
    """utils/operation_helpers.rs:5-8: keccak256 of the username bytes as a big-endian integer"""
    return int.from_bytes(keccak256(username.encode()), "big")


def big_uint_to_fp(big_uint: int) -> bytes:
    """utils/operation_helpers.rs:10-12 (`Fp::from_str_vartime` of the decimal string): the integer reduced modulo r, in
    halo2curves' memory form (32 bytes, Montgomery, little-endian) -- what every `_dev` entry point takes"""
    if big_uint < 0:
        raise ValueError("Invalid balance")
    return _to_fr_bytes(big_uint)


def fp_to_big_uint(f) -> int:
    """utils/operation_helpers.rs:15-17: the canonical integer of a field element given in memory form"""
    v = int.from_bytes(bytes(f), "little")
    if v >= R_MODULUS:
        raise ValueError("not a field element")
    return v * pow(1 << 256, -1, R_MODULUS) % R_MODULUS


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
    cryptocurrencies = []
    for h in header[1:]:            # csv_parser.rs:17-31: `balance_<name>_<chain>`, anything else is an error
        parts = h.split("_")
        if len(parts) != 3 or parts[0] != "balance":
            raise ValueError(f"Invalid header: {h}")
        cryptocurrencies.append((parts[1], parts[2]))
    if header[0] != "username":
        raise ValueError("Username not found")
    def balance(text: str) -> int:
        # BigUint::parse_bytes(.., 10): decimal digits only -- no sign, blanks or underscores (csv_parser.rs:45-50)
        if not text or not text.isascii() or not text.isdigit():
            raise ValueError("Invalid balance")
        return int(text)
    entries = [(r[0], [balance(x) for x in r[1:1 + n_currencies]]) for r in body if r]
    return entries, cryptocurrencies


def _hash_batch(kind: str, *arrays, n: int, nc: int):
    """one device call over n nodes: kind 'leaf' (usernames, balances) -> hashes; 'middle' (child hashes, child
    balances of 2n children) -> (hashes, balances)"""
    import torch
    L = ffi.lib()
    dev = [torch.from_numpy(np.array(a, dtype=np.uint8, copy=True)).cuda() for a in arrays]
    h = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    if kind == "leaf":
        ffi.check(L.sg_mst_leaves_dev(ffi.dev_ptr(dev[0]), ffi.dev_ptr(dev[1]), C.c_size_t(n), C.c_uint32(nc), ffi.dev_ptr(h),
                                      ffi.current_stream_ptr()))
        return h.cpu().numpy()
    b = torch.empty(32 * n * nc, dtype=torch.uint8, device="cuda")
    ffi.check(L.sg_mst_level_dev(ffi.dev_ptr(dev[0]), ffi.dev_ptr(dev[1]), C.c_size_t(n), C.c_uint32(nc), ffi.dev_ptr(h),
                                 ffi.dev_ptr(b), ffi.current_stream_ptr()))
    return h.cpu().numpy(), b.cpu().numpy()


class MerkleSumTree:
    """MerkleSumTree<N_CURRENCIES, N_BYTES> (mst.rs): `from_csv`, `from_csv_sorted`, `from_entries`, `from_params`, `root`,
    `leaves`, `entries`, `index_of_username`, `update_leaf`; Tree trait (tree.rs): `depth`, `nodes`, `cryptocurrencies`,
    `get_entry`, the two preimage getters, `generate_proof`, `verify_proof`; `Entry::compute_leaf / recompute_leaf` by index.  All Poseidon hashing runs on the device; nodes are kept level-major on the host."""

    def __init__(self, depth, n_currencies, entries, node_hashes, node_balances, is_sorted=False, cryptocurrencies=None):
        self.depth, self.n_currencies, self.entries = depth, n_currencies, entries
        self._h, self._b = node_hashes, node_balances  # level-major numpy buffers
        self.is_sorted = is_sorted
        self.cryptocurrencies = cryptocurrencies or []

    @classmethod
    def from_csv(cls, path: str, n_currencies: int, n_bytes: int = 8):
        entries, crypto = parse_csv_to_entries(path, n_currencies)
        t = cls.from_entries(entries, n_currencies, n_bytes)
        t.cryptocurrencies = crypto
        return t

    @classmethod
    def from_csv_sorted(cls, path: str, n_currencies: int, n_bytes: int = 8):
        """mst.rs:89-100: leaves sorted by the username byte values"""
        entries, crypto = parse_csv_to_entries(path, n_currencies)
        entries.sort(key=lambda e: e[0].encode())
        t = cls.from_entries(entries, n_currencies, n_bytes, is_sorted=True)
        t.cryptocurrencies = crypto
        return t

    @staticmethod
    def _entry_fields(name: str, bal):
        """(username field element, balance field elements) as Montgomery bytes; the zero entry is ("0", 0...)
        with username field element 0 (entry.rs:30-38)"""
        u = 0 if name is None else big_intify_username(name)
        return _to_fr_bytes(u), b"".join(_to_fr_bytes(v) for v in bal)

    @classmethod
    def from_entries(cls, entries, n_currencies: int, n_bytes: int = 8, is_sorted: bool = False):
        import torch
        n = len(entries)
        if n == 0:
            raise ValueError("empty tree")
        for _, bal in entries:
            if len(bal) != n_currencies:
                raise ValueError("entry with a wrong number of balances")
            # mst.rs:103-134 builds the tree from any BigUint balances: one that does not fit N_BYTES only fails later, in
            # the circuit's range check (circuits/tests.rs:268-299 builds its tree from entry_16_overflow.csv)
            if any(b < 0 for b in bal):
                raise ValueError("Invalid balance")
        depth = max(0, (n - 1).bit_length())
        size = 1 << depth
        users = bytearray(32 * size)      # zero entries: username 0, balances 0 (entry.rs:30-38)
        bals = bytearray(32 * size * n_currencies)
        for i, (name, bal) in enumerate(entries):
            u, b = cls._entry_fields(name, bal)
            users[32 * i:32 * i + 32] = u
            bals[32 * i * n_currencies:32 * (i + 1) * n_currencies] = b
        d_users = torch.from_numpy(np.frombuffer(bytes(users), dtype=np.uint8).copy()).cuda()
        d_bals = torch.from_numpy(np.frombuffer(bytes(bals), dtype=np.uint8).copy()).cuda()
        nodes = 2 * size - 1
        d_h = torch.empty(32 * nodes, dtype=torch.uint8, device="cuda")
        d_b = torch.empty(32 * nodes * n_currencies, dtype=torch.uint8, device="cuda")
        ffi.check(ffi.lib().sg_mst_build_dev(ffi.dev_ptr(d_users), ffi.dev_ptr(d_bals), C.c_uint32(depth),
                                             C.c_uint32(n_currencies), ffi.dev_ptr(d_h), ffi.dev_ptr(d_b),
                                             ffi.current_stream_ptr()))
        torch.cuda.synchronize()
        padded = list(entries) + [(None, [0] * n_currencies)] * (size - n)
        return cls(depth, n_currencies, padded, d_h.cpu().numpy(), d_b.cpu().numpy(), is_sorted)

    @classmethod
    def from_params(cls, root, nodes, depth: int, entries, cryptocurrencies, is_sorted: bool, n_currencies: int | None = None):
        """mst.rs:137-159: a tree from parts computed elsewhere -- `nodes[level][index]` = (hash, balances) in memory
        form, level 0 the leaves, `root` = (hash, balances).  The reference stores what it is given; here the shape is
        checked as well (2^(depth - level) nodes per level, the root equal to the top node), because the level-major
        buffers below have no room for a ragged tree."""
        if len(nodes) != depth + 1 or any(len(nodes[l]) != 1 << (depth - l) for l in range(depth + 1)):
            raise ValueError("nodes: expected 2^(depth - level) nodes on every level")
        nc = n_currencies if n_currencies is not None else len(bytes(nodes[0][0][1])) // 32
        hs = np.frombuffer(b"".join(bytes(h) for lvl in nodes for h, _ in lvl), dtype=np.uint8).copy()
        bs = np.frombuffer(b"".join(bytes(b) for lvl in nodes for _, b in lvl), dtype=np.uint8).copy()
        total = (2 << depth) - 1
        if hs.size != 32 * total or bs.size != 32 * nc * total:
            raise ValueError("nodes: every node is a 32-byte hash and n_currencies 32-byte balances")
        if bytes(root[0]) != bytes(nodes[depth][0][0]) or bytes(root[1]) != bytes(nodes[depth][0][1]):
            raise ValueError("root differs from the top node")
        size = 1 << depth
        padded = list(entries) + [(None, [0] * nc)] * (size - len(entries))
        return cls(depth, nc, padded, hs, bs, is_sorted, list(cryptocurrencies))

    def nodes(self):
        """tree.rs:15 `nodes()`: every level, leaves first, as lists of (hash, balances) views"""
        return [[self.node(level, i) for i in range(1 << (self.depth - level))] for level in range(self.depth + 1)]

    def compute_leaf(self, index: int):
        """entry.rs:40-47 `Entry::compute_leaf` of entry `index`: (hash, balances), hashed on the device"""
        return self.recompute_leaf(index, self.entries[index][1])

    def recompute_leaf(self, index: int, updated_balances):
        """entry.rs:50-58 `Entry::recompute_leaf`: the leaf this entry would have with other balances (nothing is stored)"""
        if len(updated_balances) != self.n_currencies:
            raise ValueError("wrong number of balances")
        u, b = self._entry_fields(self.entries[index][0], updated_balances)
        bb = np.frombuffer(b, dtype=np.uint8)
        return _hash_batch("leaf", np.frombuffer(u, dtype=np.uint8), bb, n=1, nc=self.n_currencies), bb.copy()

    def _level_offset(self, level: int) -> int:
        size = 1 << self.depth
        return sum(size >> l for l in range(level))

    def node(self, level: int, index: int):
        o = self._level_offset(level) + index
        nc = self.n_currencies
        return self._h[32 * o:32 * o + 32], self._b[32 * o * nc:32 * (o + 1) * nc]

    def _set_node(self, level: int, index: int, h, b):
        o = self._level_offset(level) + index
        nc = self.n_currencies
        self._h[32 * o:32 * o + 32] = h
        self._b[32 * o * nc:32 * (o + 1) * nc] = b

    def root(self):
        return self.node(self.depth, 0)

    def leaves(self):
        """(hashes, balances) of level 0"""
        size = 1 << self.depth
        return self._h[:32 * size], self._b[:32 * size * self.n_currencies]

    def index_of_username(self, username: str) -> int:
        """mst.rs:207-223: linear scan, or binary search when the tree was built sorted"""
        if self.is_sorted:
            real = [e for e in self.entries if e[0] is not None]
            lo, hi = 0, len(real)
            key = username.encode()
            while lo < hi:
                mid = (lo + hi) // 2
                if real[mid][0].encode() < key:
                    lo = mid + 1
                else:
                    hi = mid
            if lo < len(real) and real[lo][0] == username:
                return lo
            raise KeyError("Username not found")
        for i, (name, _) in enumerate(self.entries):
            if name == username:
                return i
        raise KeyError("Username not found")

    def update_leaf(self, username: str, new_balances):
        """mst.rs:169-204: new balances for one user, the leaf and its `depth` ancestors are re-hashed (device,
        one call per level); returns the new root (hash, balances)"""
        nc = self.n_currencies
        if len(new_balances) != nc:
            raise ValueError("wrong number of balances")
        if any(int(b) < 0 for b in new_balances):
            raise ValueError("Invalid balance")
        index = self.index_of_username(username)
        self.entries[index] = (username, list(new_balances))
        u, b = self._entry_fields(username, new_balances)
        ub, bb = np.frombuffer(u, dtype=np.uint8), np.frombuffer(b, dtype=np.uint8)
        self._set_node(0, index, _hash_batch("leaf", ub, bb, n=1, nc=nc), bb)
        cur = index
        for level in range(1, self.depth + 1):
            parent = cur // 2
            lh, lb = self.node(level - 1, 2 * parent)
            rh, rb = self.node(level - 1, 2 * parent + 1)
            h, bsum = _hash_batch("middle", np.concatenate([lh, rh]), np.concatenate([lb, rb]), n=1, nc=nc)
            self._set_node(level, parent, h, bsum)
            cur = parent
        return self.root()

    def get_entry(self, index: int):
        return self.entries[index]

    def get_middle_node_hash_preimage(self, level: int, index: int) -> np.ndarray:
        """tree.rs:22-56: [left.balances + right.balances ..., left.hash, right.hash] of the middle node (level, index) as
        Montgomery bytes; "Invalid depth" for the leaf level or above the root, "Node not found" beyond the level's width"""
        if level == 0 or level > self.depth:
            raise ValueError("Invalid depth")
        if not 0 <= index < (1 << (self.depth - level)):
            raise IndexError("Node not found")
        lh, _ = self.node(level - 1, 2 * index)
        rh, _ = self.node(level - 1, 2 * index + 1)
        return np.concatenate([self.node(level, index)[1], lh, rh])

    def get_leaf_node_hash_preimage(self, index: int) -> np.ndarray:
        """tree.rs:59-81: [username, balances ...] of entry `index` as Montgomery bytes"""
        name, bal = self.get_entry(index)
        u, b = self._entry_fields(name, bal)
        return np.frombuffer(u + b, dtype=np.uint8).copy()

    @staticmethod
    def middle_node_from_preimage(preimage, n_currencies: int):
        """node.rs:73-84 `Node::middle_node_from_preimage`: (hash, balances) with hash = H(preimage) on the device"""
        pre = np.asarray(preimage, dtype=np.uint8)
        bal = pre[:32 * n_currencies]
        zeros = np.zeros(32 * n_currencies, dtype=np.uint8)
        h, _ = _hash_batch("middle", pre[32 * n_currencies:], np.concatenate([bal, zeros]), n=1, nc=n_currencies)
        return h, bal.copy()

    @staticmethod
    def leaf_node_from_preimage(preimage, n_currencies: int):
        """node.rs:57-68 `Node::leaf_node_from_preimage`"""
        pre = np.asarray(preimage, dtype=np.uint8)
        return _hash_batch("leaf", pre[:32], pre[32:], n=1, nc=n_currencies), pre[32:].copy()

    def generate_proof(self, index: int):
        """tree.rs:85-137.  Besides the sibling nodes (hash, balances) the proof carries what the reference's
        MerkleProof holds: the sibling leaf's hash preimage [username, balances...] and, for every level above,
        the sibling middle node's preimage [left.bal + right.bal ..., left.hash, right.hash]."""
        if not 0 <= index < (1 << self.depth):
            raise IndexError("Index out of bounds")
        nc = self.n_currencies
        sib, bits, pre_mid, idx = [], [], [], index
        for level in range(self.depth):
            bits.append(idx & 1)
            sidx = idx ^ 1
            sib.append(self.node(level, sidx))
            if level > 0:
                lh, _ = self.node(level - 1, 2 * sidx)
                rh, _ = self.node(level - 1, 2 * sidx + 1)
                pre_mid.append(np.concatenate([self.node(level, sidx)[1], lh, rh]))
            idx >>= 1
        pre_leaf = None
        if self.depth:
            name, bal = self.entries[index ^ 1]
            u, b = self._entry_fields(name, bal)
            pre_leaf = np.frombuffer(u + b, dtype=np.uint8).copy()
        return {"entry": self.entries[index], "leaf": self.node(0, index), "siblings": sib, "path_indices": bits,
                "root": self.root(), "sibling_leaf_node_hash_preimage": pre_leaf,
                "sibling_middle_node_hash_preimages": pre_mid}

    def verify_proof(self, proof) -> bool:
        """tree.rs:140-190: recompute the path from the entry and the sibling preimages (hashes on the device),
        compare the root hash and balances"""
        nc = self.n_currencies
        name, bal = proof["entry"]
        u, b = self._entry_fields(name, bal)
        bb = np.frombuffer(b, dtype=np.uint8)
        node_h, node_b = _hash_batch("leaf", np.frombuffer(u, dtype=np.uint8), bb, n=1, nc=nc), bb
        for level, bit in enumerate(proof["path_indices"]):
            if level == 0:
                pre = proof["sibling_leaf_node_hash_preimage"]
                sib_b = pre[32:]
                sib_h = _hash_batch("leaf", pre[:32], sib_b, n=1, nc=nc)
            else:
                pre = proof["sibling_middle_node_hash_preimages"][level - 1]
                sib_b = pre[:32 * nc]
                # a middle node is H(balances..., left.hash, right.hash): rebuild it from two pseudo-children whose
                # balances add up to the preimage's (left = the sums, right = zero)
                zeros = np.zeros(32 * nc, dtype=np.uint8)
                sib_h, chk_b = _hash_batch("middle", pre[32 * nc:], np.concatenate([sib_b, zeros]), n=1, nc=nc)
                if not (chk_b == sib_b).all():
                    return False
            pair_h = np.concatenate([sib_h, node_h]) if bit else np.concatenate([node_h, sib_h])
            pair_b = np.concatenate([sib_b, node_b]) if bit else np.concatenate([node_b, sib_b])
            node_h, node_b = _hash_batch("middle", pair_h, pair_b, n=1, nc=nc)
        rh, rb = proof["root"]
        return bool((node_h == rh).all() and (node_b == rb).all())


class DeviceMerkleSumTree:
    """A Merkle sum tree whose leaves are already field elements on the device (a snapshot of 2^depth users: username
    field elements and balances as Montgomery Fr), built and KEPT in HBM: level-major node arrays as `sg_mst_build_dev`
    lays them out.  The reference's `Tree` trait [REF zk_prover/src/merkle_sum_tree/tree.rs:8-137] for the two methods a
    prover needs -- `root`, `generate_proof` --; a Merkle proof costs one gather of 3 * depth + 2 rows and one small
    device-to-host copy.  (MerkleSumTree above is the CSV / entries flavour with the nodes mirrored on the host.)"""

    def __init__(self, d_usernames, d_balances, depth: int, n_currencies: int):
        import torch
        size = 1 << depth
        if d_usernames.numel() != 32 * size or d_balances.numel() != 32 * size * n_currencies:
            raise ValueError("DeviceMerkleSumTree: 2^depth usernames and 2^depth * n_currencies balances expected")
        self.depth, self.n_currencies = depth, n_currencies
        self.d_users, self.d_bals = d_usernames, d_balances
        nodes = 2 * size - 1
        self.d_h = torch.empty(32 * nodes, dtype=torch.uint8, device="cuda")
        self.d_b = torch.empty(32 * nodes * n_currencies, dtype=torch.uint8, device="cuda")
        ffi.check(ffi.lib().sg_mst_build_dev(ffi.dev_ptr(d_usernames), ffi.dev_ptr(d_balances), C.c_uint32(depth),
                                             C.c_uint32(n_currencies), ffi.dev_ptr(self.d_h), ffi.dev_ptr(self.d_b),
                                             ffi.current_stream_ptr()))
        # the snapshot is read from other streams afterwards (proofs in flight run one stream per worker thread, and the
        # first reader caches the root): the build -- one-off, tens of milliseconds -- is complete when the constructor returns
        torch.cuda.current_stream().synchronize()
        self.offsets = [0]
        for level in range(depth):
            self.offsets.append(self.offsets[-1] + (size >> level))
        self._root = None

    def _rows(self, hash_nodes, balance_nodes, users):
        """one gather + one copy: rows of the hash array, of the balance array (n_currencies rows per node), leaf inputs"""
        import torch
        nc = self.n_currencies
        dev = lambda idx: torch.tensor(idx, dtype=torch.int64, device="cuda")
        parts = [self.d_h.view(-1, 32)[dev(hash_nodes)].reshape(-1) if hash_nodes else None,
                 self.d_b.view(-1, 32 * nc)[dev(balance_nodes)].reshape(-1) if balance_nodes else None,
                 self.d_users.view(-1, 32)[dev(users)].reshape(-1) if users else None,
                 self.d_bals.view(-1, 32 * nc)[dev(users)].reshape(-1) if users else None]
        flat = torch.cat([p for p in parts if p is not None]).cpu().numpy()
        out, pos = [], 0
        for p, width in zip(parts, (32, 32 * nc, 32, 32 * nc)):
            count = 0 if p is None else p.numel() // width
            out.append([flat[pos + width * i:pos + width * (i + 1)] for i in range(count)])
            pos += width * count
        return out

    def root(self):
        if self._root is None:
            h, b, _, _ = self._rows([self.offsets[self.depth]], [self.offsets[self.depth]], [])
            self._root = (h[0], b[0])
        return self._root

    def public_inputs(self, index: int):
        """[leaf hash, root hash, root balances..] of user `index` as integers: `circuit.instances()[0]`"""
        if not 0 <= index < (1 << self.depth):
            raise IndexError("Index out of bounds")
        h, _, _, _ = self._rows([index], [], [])
        rinv = pow(1 << 256, -1, R_MODULUS)
        to_int = lambda row: int.from_bytes(bytes(row), "little") * rinv % R_MODULUS
        rh, rb = self.root()
        return [to_int(h[0]), to_int(rh)] + [to_int(rb[32 * c:32 * c + 32]) for c in range(self.n_currencies)]

    def public_inputs_many(self, indices):
        """`public_inputs` of several users with one gather and one copy"""
        if any(not 0 <= i < (1 << self.depth) for i in indices):
            raise IndexError("Index out of bounds")
        h, _, _, _ = self._rows([int(i) for i in indices], [], [])
        rinv = pow(1 << 256, -1, R_MODULUS)
        to_int = lambda row: int.from_bytes(bytes(row), "little") * rinv % R_MODULUS
        rh, rb = self.root()
        tail = [to_int(rh)] + [to_int(rb[32 * c:32 * c + 32]) for c in range(self.n_currencies)]
        return [[to_int(row)] + tail for row in h]

    def generate_proof(self, index: int):
        """the fields of the reference's MerkleProof (see MerkleSumTree.generate_proof); `entry` carries the username as
        its field element (int), since a device snapshot holds no names"""
        if not 0 <= index < (1 << self.depth):
            raise IndexError("Index out of bounds")
        hash_nodes, bal_nodes, bits = [self.offsets[0] + index], [], []
        idx = index
        for level in range(self.depth):
            bits.append(idx & 1)
            sib = idx ^ 1
            if level > 0:   # the sibling middle node's preimage: its balances, its children's hashes
                bal_nodes.append(self.offsets[level] + sib)
                hash_nodes += [self.offsets[level - 1] + 2 * sib, self.offsets[level - 1] + 2 * sib + 1]
            idx >>= 1
        h, b, users, bals = self._rows(hash_nodes, bal_nodes, [index, index ^ 1] if self.depth else [index])
        rinv = pow(1 << 256, -1, R_MODULUS)
        to_int = lambda row: int.from_bytes(bytes(row), "little") * rinv % R_MODULUS
        nc = self.n_currencies
        entry = (to_int(users[0]), [to_int(bals[0][32 * c:32 * c + 32]) for c in range(nc)])
        pre_mid = [np.concatenate([b[l], h[1 + 2 * l], h[2 + 2 * l]]) for l in range(max(0, self.depth - 1))]
        pre_leaf = np.concatenate([users[1], bals[1]]) if self.depth else None
        return {"entry": entry, "leaf": (h[0], bals[0]), "path_indices": bits, "root": self.root(),
                "sibling_leaf_node_hash_preimage": pre_leaf, "sibling_middle_node_hash_preimages": pre_mid}
