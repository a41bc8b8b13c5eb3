"""
oracle.huffman_tables -- the four trained Huffman code tables as plain data (TEST ORACLE).

Contents transcribed from the reference's training_data/{percussive,silence,speech,tonal}_table.pkl
(protocol-0 pickles read AS TEXT; nothing is unpickled) -- see SURVEY.md A.3 and
tools/check_huffman_tables.py, which re-derives them from the pickle text when the reference is
present.  value -> (code string, code length); the escape value is listed last.

The reference indexes tables by os.walk/glob directory order, which is filesystem dependent and
differs between its encoder (`*table.pkl`) and decoder (`*tree.pkl`, `*.revpkl`)
(codecThem.py:137-138, pacfileThem.py:170-171,246).  This build fixes ONE order -- sorted names --
in the oracle and in the product; "bit-identical" Huffman output is defined against that order.
"""

RAW_TABLE_ID = 15          # codecThem.py:149 -- 4-bit table id meaning "mantissas stored raw"
TABLE_ORDER = ("percussive", "silence", "speech", "tonal")


def _t(pairs):
    return {v: (c, len(c)) for v, c in pairs}


TABLES = {
    "percussive": (_t([(0, "0"), (1, "1110"), (2, "110"), (3, "111101"), (4, "101"), (5, "1000"),
                       (6, "111100"), (7, "11111100"), (8, "10010"), (9, "111110"), (10, "1111111"),
                       (11, "1001101"), (12, "1001100"), (13, "111111011"), (14, "111111010"),
                       (16, "100111")]), 16),
    "silence": (_t([(0, "11"), (1, "000"), (2, "100"), (3, "00101"), (4, "01"), (5, "0011"),
                    (6, "101101"), (8, "10111"), (9, "00100"), (10, "101100"),
                    (11, "1010")]), 11),
    "speech": (_t([(0, "11"), (1, "1001"), (2, "101"), (3, "100011"), (4, "00"), (5, "0100"),
                   (6, "100000"), (8, "0111"), (9, "01101"), (10, "100001"), (11, "1000101"),
                   (12, "0110000"), (16, "011001"), (17, "1000100"), (32, "0110001"),
                   (7, "0101")]), 7),
    "tonal": (_t([(0, "0"), (1, "11110"), (2, "110"), (3, "1111101"), (4, "101"), (5, "1001011"),
                  (6, "11111101"), (8, "1110"), (9, "1111111"), (10, "10010100"), (16, "10011"),
                  (17, "1111100"), (18, "11111100"), (32, "100100"), (64, "10010101"),
                  (7, "1000")]), 7),
}
