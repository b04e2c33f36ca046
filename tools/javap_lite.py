#!/usr/bin/env python3
"""javap_lite — minimal JVM class-file disassembler (constant pool + method bytecode).

The image has no JDK, and third-party arithmetic on the hot path (cc.mallet:mallet:2.0.8) exists
in the reference only as class files inside output/lib/mallet-2.0.8.jar.  This tool prints a
method's bytecode so that its published algorithm can be restated exactly in the oracle / host
code.  Study tool only: it reads a class file, it copies nothing.

  python tools/javap_lite.py <jar> <class/path/Name.class> [method-name ...]
"""
import struct
import sys
import zipfile

OPC = {}
def _ops():
    names = """nop aconst_null iconst_m1 iconst_0 iconst_1 iconst_2 iconst_3 iconst_4 iconst_5 lconst_0 lconst_1
fconst_0 fconst_1 fconst_2 dconst_0 dconst_1 bipush sipush ldc ldc_w ldc2_w iload lload fload dload aload
iload_0 iload_1 iload_2 iload_3 lload_0 lload_1 lload_2 lload_3 fload_0 fload_1 fload_2 fload_3
dload_0 dload_1 dload_2 dload_3 aload_0 aload_1 aload_2 aload_3 iaload laload faload daload aaload baload caload saload
istore lstore fstore dstore astore istore_0 istore_1 istore_2 istore_3 lstore_0 lstore_1 lstore_2 lstore_3
fstore_0 fstore_1 fstore_2 fstore_3 dstore_0 dstore_1 dstore_2 dstore_3 astore_0 astore_1 astore_2 astore_3
iastore lastore fastore dastore aastore bastore castore sastore pop pop2 dup dup_x1 dup_x2 dup2 dup2_x1 dup2_x2 swap
iadd ladd fadd dadd isub lsub fsub dsub imul lmul fmul dmul idiv ldiv fdiv ddiv irem lrem frem drem
ineg lneg fneg dneg ishl lshl ishr lshr iushr lushr iand land ior lor ixor lxor iinc
i2l i2f i2d l2i l2f l2d f2i f2l f2d d2i d2l d2f i2b i2c i2s lcmp fcmpl fcmpg dcmpl dcmpg
ifeq ifne iflt ifge ifgt ifle if_icmpeq if_icmpne if_icmplt if_icmpge if_icmpgt if_icmple if_acmpeq if_acmpne
goto jsr ret tableswitch lookupswitch ireturn lreturn freturn dreturn areturn return
getstatic putstatic getfield putfield invokevirtual invokespecial invokestatic invokeinterface invokedynamic
new newarray anewarray arraylength athrow checkcast instanceof monitorenter monitorexit wide multianewarray ifnull ifnonnull goto_w jsr_w""".split()
    for i, n in enumerate(names):
        OPC[i] = n
_ops()
ONE = {"bipush", "ldc", "iload", "lload", "fload", "dload", "aload", "istore", "lstore", "fstore", "dstore", "astore", "ret", "newarray"}
TWO = {"sipush", "ldc_w", "ldc2_w", "getstatic", "putstatic", "getfield", "putfield", "invokevirtual", "invokespecial",
       "invokestatic", "new", "anewarray", "checkcast", "instanceof"}
BR = {"ifeq", "ifne", "iflt", "ifge", "ifgt", "ifle", "if_icmpeq", "if_icmpne", "if_icmplt", "if_icmpge", "if_icmpgt",
      "if_icmple", "if_acmpeq", "if_acmpne", "goto", "jsr", "ifnull", "ifnonnull"}


def parse(data):
    pos = 8
    n = struct.unpack(">H", data[pos:pos + 2])[0]; pos += 2
    cp = [None] * n
    i = 1
    while i < n:
        t = data[pos]; pos += 1
        if t == 1:
            l = struct.unpack(">H", data[pos:pos + 2])[0]; pos += 2
            cp[i] = ("utf8", data[pos:pos + l].decode("utf8", "replace")); pos += l
        elif t in (3, 4):
            v = struct.unpack(">i" if t == 3 else ">f", data[pos:pos + 4])[0]; cp[i] = ("num", v); pos += 4
        elif t in (5, 6):
            v = struct.unpack(">q" if t == 5 else ">d", data[pos:pos + 8])[0]; cp[i] = ("num", v); pos += 8; i += 1
        elif t in (7, 8, 16, 19, 20):
            cp[i] = ("ref1", t, struct.unpack(">H", data[pos:pos + 2])[0]); pos += 2
        elif t in (9, 10, 11, 12, 17, 18):
            cp[i] = ("ref2", t, struct.unpack(">HH", data[pos:pos + 4])); pos += 4
        elif t == 15:
            cp[i] = ("mh", data[pos], struct.unpack(">H", data[pos + 1:pos + 3])[0]); pos += 3
        else:
            raise ValueError(f"cp tag {t}")
        i += 1

    def cstr(k):
        e = cp[k]
        if e is None: return "?"
        if e[0] == "utf8": return e[1]
        if e[0] == "num": return repr(e[1])
        if e[0] == "ref1": return cstr(e[2])
        if e[0] == "ref2": return cstr(e[2][0]) + "." + cstr(e[2][1]) if e[1] != 12 else cstr(e[2][0]) + ":" + cstr(e[2][1])
        return str(e)

    pos += 6
    ni = struct.unpack(">H", data[pos:pos + 2])[0]; pos += 2 + 2 * ni

    def skip_attrs(pos, want_code=False):
        na = struct.unpack(">H", data[pos:pos + 2])[0]; pos += 2
        code = None
        for _ in range(na):
            nm, ln = struct.unpack(">HI", data[pos:pos + 6]); pos += 6
            if want_code and cstr(nm) == "Code":
                cl = struct.unpack(">I", data[pos + 4:pos + 8])[0]
                code = data[pos + 8:pos + 8 + cl]
            pos += ln
        return pos, code

    nf = struct.unpack(">H", data[pos:pos + 2])[0]; pos += 2
    for _ in range(nf):
        pos += 6
        pos, _c = skip_attrs(pos)
    nm = struct.unpack(">H", data[pos:pos + 2])[0]; pos += 2
    methods = []
    for _ in range(nm):
        _acc, name, desc = struct.unpack(">HHH", data[pos:pos + 6]); pos += 6
        pos, code = skip_attrs(pos, True)
        methods.append((cstr(name), cstr(desc), code))
    return cstr, methods


def disasm(code, cstr):
    pc = 0
    out = []
    while pc < len(code):
        op = code[pc]; name = OPC.get(op, f"op{op}")
        if name in ONE:
            arg = code[pc + 1]
            s = f"{name} {cstr(arg) if name == 'ldc' else (struct.unpack('b', bytes([arg]))[0] if name == 'bipush' else arg)}"; ln = 2
        elif name in TWO:
            k = struct.unpack(">H", code[pc + 1:pc + 3])[0]
            s = f"{name} {struct.unpack('>h', code[pc+1:pc+3])[0] if name == 'sipush' else cstr(k)}"; ln = 3
        elif name in BR:
            off = struct.unpack(">h", code[pc + 1:pc + 3])[0]; s = f"{name} -> {pc + off}"; ln = 3
        elif name == "iinc":
            s = f"iinc {code[pc+1]} {struct.unpack('b', bytes([code[pc+2]]))[0]}"; ln = 3
        elif name == "invokeinterface" or name == "invokedynamic":
            s = f"{name} {cstr(struct.unpack('>H', code[pc+1:pc+3])[0])}"; ln = 5
        elif name == "multianewarray":
            s = name; ln = 4
        elif name in ("tableswitch", "lookupswitch"):
            pad = (4 - (pc + 1) % 4) % 4; p = pc + 1 + pad
            if name == "tableswitch":
                d, lo, hi = struct.unpack(">iii", code[p:p + 12]); ln = 1 + pad + 12 + 4 * (hi - lo + 1)
            else:
                d, npairs = struct.unpack(">ii", code[p:p + 8]); ln = 1 + pad + 8 + 8 * npairs
            s = name
        elif name == "wide":
            s = "wide"; ln = 6 if OPC.get(code[pc + 1]) == "iinc" else 4
        else:
            s = name; ln = 1
        out.append(f"{pc:5d}: {s}")
        pc += ln
    return out


if __name__ == "__main__":
    jar, cls = sys.argv[1], sys.argv[2]
    want = set(sys.argv[3:])
    data = zipfile.ZipFile(jar).read(cls)
    cstr, methods = parse(data)
    for name, desc, code in methods:
        if want and name not in want:
            continue
        print(f"== {name}{desc}  ({0 if code is None else len(code)} bytes)")
        if code is not None:
            print("\n".join(disasm(code, cstr)))
