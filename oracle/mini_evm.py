"""A minimal EVM interpreter (TEST INFRASTRUCTURE ONLY -- nothing under ezkl_amd/ may import it).

Why it exists: the reference ships, next to its fixture proof, the deployment bytecode of a Solidity verifier its own tooling generated
for a k = 6 circuit (/root/reference/tests/assets/wasm.code -> tests/golden/verifier_k6.code, copied by tests/golden/make_golden.py).
halo2 itself is not in the reference tree, but this contract IS the zkonduit halo2 verifier for the EVM transcript, compiled: running
it on a byte string shows, as KECCAK256 inputs and calldata reads, exactly what the reference's protocol absorbs in which order, how a
challenge is squeezed, which proof offsets each evaluation is read from and how the SHPLONK / pairing inputs are formed.  Tests trace
it and compare with ezkl_amd's transcript and proof layout (tests/test_evm_verifier.py).

Only what solc 0.8.20 emitted for that contract is implemented: the stack / memory / calldata / control-flow opcodes, KECCAK256 and
STATICCALL to the precompiles 5 (modexp), 6 (bn254 add), 7 (bn254 mul) and 8 (bn254 pairing check)."""
from ezkl_amd.transcript import keccak256
from oracle import pairing as E

M256 = (1 << 256) - 1
Q = E.Q


def _signed(x): return x - (1 << 256) if x >> 255 else x


class Revert(Exception):
    def __init__(self, data): self.data = data


class Evm:
    def __init__(self, code, calldata, trace=None):
        self.code, self.calldata = bytes(code), bytes(calldata)
        self.mem = bytearray()
        self.stack = []
        self.ret = b""                       # last returndata
        self.trace = trace if trace is not None else {}
        self.trace.setdefault("keccak", [])   # inputs of every KECCAK256
        self.trace.setdefault("calldataload", [])
        self.trace.setdefault("calls", [])    # (precompile, input bytes, output bytes)
        self.on_addmod = self.on_mulmod = None   # optional hooks (a, b, result): tests follow the quotient accumulation with them
        self.jumpdests = set()
        i = 0
        while i < len(self.code):
            op = self.code[i]
            if op == 0x5b: self.jumpdests.add(i)
            i += 1 + (op - 0x5f if 0x60 <= op <= 0x7f else 0)

    # ---- memory
    def _grow(self, end):
        if end > len(self.mem):
            self.mem.extend(b"\0" * ((end + 31) // 32 * 32 - len(self.mem)))

    def mload(self, off, size=32):
        if size == 0: return b""
        self._grow(off + size)
        return bytes(self.mem[off:off + size])

    def mstore(self, off, data):
        if not data: return
        self._grow(off + len(data))
        self.mem[off:off + len(data)] = data

    # ---- precompiles
    def precompile(self, addr, data):
        if addr == 5:
            bl, el, ml = (int.from_bytes(data[i:i + 32].ljust(32, b"\0"), "big") for i in (0, 32, 64))
            body = data[96:].ljust(bl + el + ml, b"\0")
            b, e, m = int.from_bytes(body[:bl], "big"), int.from_bytes(body[bl:bl + el], "big"), int.from_bytes(body[bl + el:bl + el + ml], "big")
            return True, (pow(b, e, m) if m else 0).to_bytes(ml, "big")
        def g1(b):
            x, y = int.from_bytes(b[:32], "big"), int.from_bytes(b[32:64], "big")
            if x >= Q or y >= Q: raise ValueError
            if x == 0 and y == 0: return None
            if (y * y - x * x * x - 3) % Q: raise ValueError
            return (x, y)
        def enc(p): return b"\0" * 64 if p is None else p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")
        try:
            if addr == 6:
                data = data.ljust(128, b"\0")
                return True, enc(E.g1_add(g1(data[:64]), g1(data[64:128])))
            if addr == 7:
                data = data.ljust(96, b"\0")
                return True, enc(E.g1_mul(g1(data[:64]), int.from_bytes(data[64:96], "big")))
            if addr == 8:
                if len(data) % 192: return False, b""
                pairs = []
                for i in range(0, len(data), 192):
                    p = g1(data[i:i + 64])
                    v = [int.from_bytes(data[i + 64 + 32 * j:i + 96 + 32 * j], "big") for j in range(4)]
                    g2 = ((v[1], v[0]), (v[3], v[2]))          # EIP-197: imaginary part first
                    if p is not None: pairs.append((p, g2))
                return True, (1 if E.pairing_check(pairs) else 0).to_bytes(32, "big")
        except ValueError:
            return False, b""
        raise NotImplementedError("precompile %d" % addr)

    # ---- run
    def run(self, max_steps=50_000_000):
        code, st, pc = self.code, self.stack, 0
        pop, push = st.pop, st.append
        for _ in range(max_steps):
            op = code[pc] if pc < len(code) else 0
            pc += 1
            if 0x60 <= op <= 0x7f:
                nb = op - 0x5f
                push(int.from_bytes(code[pc:pc + nb], "big")); pc += nb
            elif op == 0x5f: push(0)
            elif 0x80 <= op <= 0x8f: push(st[-(op - 0x7f)])
            elif 0x90 <= op <= 0x9f:
                k = op - 0x8f
                st[-1], st[-1 - k] = st[-1 - k], st[-1]
            elif op == 0x50: pop()
            elif op == 0x01: push((pop() + pop()) & M256)
            elif op == 0x02: push((pop() * pop()) & M256)
            elif op == 0x03: a, b = pop(), pop(); push((a - b) & M256)
            elif op == 0x04: a, b = pop(), pop(); push(a // b if b else 0)
            elif op == 0x06: a, b = pop(), pop(); push(a % b if b else 0)
            elif op == 0x08:
                a, b, n = pop(), pop(), pop(); push((a + b) % n if n else 0)
                if self.on_addmod: self.on_addmod(a, b, st[-1])
            elif op == 0x09:
                a, b, n = pop(), pop(), pop(); push((a * b) % n if n else 0)
                if self.on_mulmod: self.on_mulmod(a, b, st[-1])
            elif op == 0x0a: a, b = pop(), pop(); push(pow(a, b, 1 << 256))
            elif op == 0x10: a, b = pop(), pop(); push(int(a < b))
            elif op == 0x11: a, b = pop(), pop(); push(int(a > b))
            elif op == 0x12: a, b = pop(), pop(); push(int(_signed(a) < _signed(b)))
            elif op == 0x13: a, b = pop(), pop(); push(int(_signed(a) > _signed(b)))
            elif op == 0x14: push(int(pop() == pop()))
            elif op == 0x15: push(int(pop() == 0))
            elif op == 0x16: push(pop() & pop())
            elif op == 0x17: push(pop() | pop())
            elif op == 0x18: push(pop() ^ pop())
            elif op == 0x19: push(pop() ^ M256)
            elif op == 0x1a: i, x = pop(), pop(); push((x >> (8 * (31 - i))) & 0xff if i < 32 else 0)
            elif op == 0x1b: s, x = pop(), pop(); push((x << s) & M256 if s < 256 else 0)
            elif op == 0x1c: s, x = pop(), pop(); push(x >> s if s < 256 else 0)
            elif op == 0x20:
                off, size = pop(), pop()
                data = self.mload(off, size)
                self.trace["keccak"].append(data)
                push(int.from_bytes(keccak256(data), "big"))
            elif op == 0x34: push(0)                                  # CALLVALUE
            elif op == 0x35:
                off = pop()
                self.trace["calldataload"].append(off)
                push(int.from_bytes(self.calldata[off:off + 32].ljust(32, b"\0"), "big"))
            elif op == 0x36: push(len(self.calldata))
            elif op == 0x37:
                d, o, s = pop(), pop(), pop()
                self.mstore(d, self.calldata[o:o + s].ljust(s, b"\0"))
            elif op == 0x38: push(len(code))
            elif op == 0x39:
                d, o, s = pop(), pop(), pop()
                self.mstore(d, code[o:o + s].ljust(s, b"\0"))
            elif op == 0x3d: push(len(self.ret))
            elif op == 0x3e:
                d, o, s = pop(), pop(), pop()
                self.mstore(d, self.ret[o:o + s].ljust(s, b"\0"))
            elif op == 0x51: push(int.from_bytes(self.mload(pop()), "big"))
            elif op == 0x52: off, v = pop(), pop(); self.mstore(off, v.to_bytes(32, "big"))
            elif op == 0x53: off, v = pop(), pop(); self.mstore(off, bytes([v & 0xff]))
            elif op == 0x56:
                pc = pop()
                assert pc in self.jumpdests, "bad jump"
            elif op == 0x57:
                dst, c = pop(), pop()
                if c:
                    pc = dst
                    assert pc in self.jumpdests, "bad jump"
            elif op == 0x58: push(pc - 1)
            elif op == 0x59: push(len(self.mem))
            elif op == 0x5a: push(M256 >> 1)                          # GAS
            elif op == 0x5b: pass
            elif op == 0xfa:                                         # STATICCALL
                _gas, addr, io, isz, oo, osz = (pop() for _ in range(6))
                data = self.mload(io, isz)
                ok, out = self.precompile(addr, data)
                self.trace["calls"].append((addr, data, out if ok else None))
                self.ret = out if ok else b""
                if ok: self.mstore(oo, out[:osz])
                push(int(ok))
            elif op == 0xf3:
                off, size = pop(), pop()
                return self.mload(off, size)
            elif op == 0xfd:
                off, size = pop(), pop()
                raise Revert(self.mload(off, size))
            elif op == 0x00: return b""
            elif op == 0xfe: raise Revert(b"INVALID")
            else:
                raise NotImplementedError("opcode 0x%02x at %d" % (op, pc - 1))
        raise RuntimeError("step limit")


def runtime_of(creation_hex):
    """the runtime code a solc creation blob returns (run the constructor)"""
    code = bytes.fromhex(creation_hex.strip())
    return Evm(code, b"").run()


def verify_proof_calldata(proof, instances):
    """verifyProof(bytes proof, uint256[] instances) -- selector 0x1e8e1e13"""
    head = bytes.fromhex("1e8e1e13")
    plen = len(proof)
    pad = (-plen) % 32
    off_inst = 0x40 + 32 + plen + pad
    body = (0x40).to_bytes(32, "big") + off_inst.to_bytes(32, "big") + plen.to_bytes(32, "big") + bytes(proof) + b"\0" * pad
    body += len(instances).to_bytes(32, "big") + b"".join(int(v).to_bytes(32, "big") for v in instances)
    return head + body
