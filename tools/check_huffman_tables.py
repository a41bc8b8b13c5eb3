#!/usr/bin/env python3
"""
Re-derive the four Huffman tables from the TEXT of the reference's protocol-0 pickles and compare
with oracle/huffman_tables.py.  Nothing is unpickled: the files are tokenised as text.  Entry
layout in the text: a key -- either `I<int>` or a numpy int32 scalar spelled
`S'\\x02\\x00\\x00\\x00'` (printable bytes appear as themselves, e.g. `S' \\x00\\x00\\x00'` = 32) -- followed by `(S'<code>'`, `I<len>`, `t`, then `s` (dict setitem);
the list's second element `I<escape>` follows the dict.
Usage: python tools/check_huffman_tables.py [/root/reference/training_data]
"""
import ast, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle.huffman_tables import TABLES, TABLE_ORDER

def parse(path):
    lines = open(path, "r", encoding="latin1").read().split("\n")
    table, key, i = {}, None, 0
    # the first 4-byte string in the file is the numpy scalar for key 0 (dtype header precedes it)
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^b?S('.*')$", ln)
        if m and not re.match(r"^'[01]+'$", m.group(1)):
            lit = ast.literal_eval(m.group(1))          # string literal -> text; no code runs
            if len(lit) == 4:                            # a little-endian int32 payload (numpy scalar key)
                key = int.from_bytes(lit.encode("latin1"), "little", signed=True)
        m = re.match(r"^s?I(-?\d+)$", ln)
        if m and i + 1 < len(lines) and lines[i + 1].startswith("(S'") and key is None:
            key = int(m.group(1))
        m = re.match(r"^\(S'([01]+)'$", ln)
        if m and key is not None:
            code = m.group(1)
            ln2 = lines[i + 2]
            assert ln2 == "I%d" % len(code), (path, ln2, code)
            table[key] = (code, len(code)); key = None
        i += 1
    # escape: last `aI<int>` / `sI<int>` followed by `a.` -- take the last bare integer line
    esc = [int(m.group(1)) for m in (re.match(r"^s?a?I(-?\d+)$", l) for l in lines) if m][-1]
    return table, esc

root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/training_data"
ok = True
for name in TABLE_ORDER:
    t, esc = parse(os.path.join(root, name + "_table.pkl"))
    want, wesc = TABLES[name]
    same = (t == want and esc == wesc)
    ok &= same
    print(name, "OK" if same else "MISMATCH", len(t), "entries, escape", esc)
    if not same:
        print("  parsed:", sorted(t.items()), esc); print("  oracle:", sorted(want.items()), wesc)
sys.exit(0 if ok else 1)
