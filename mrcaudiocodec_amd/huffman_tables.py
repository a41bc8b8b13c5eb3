"""
The four trained Huffman code tables of the reference (training_data/*_table.pkl), as plain data for
the host-side Huffman stage (codecThem.py:136-203).  value -> code string; escape value given
separately.  Table ids follow the sorted-name order fixed in DESIGN.md (the reference's own order is
whatever os.walk/glob returns, codecThem.py:137-138); id 15 = raw mantissas (codecThem.py:149).
"""
RAW_TABLE_ID = 15
TABLE_NAMES = ("percussive", "silence", "speech", "tonal")

CODES = {
    "percussive": {0: "0", 1: "1110", 2: "110", 3: "111101", 4: "101", 5: "1000", 6: "111100", 7: "11111100",
                   8: "10010", 9: "111110", 10: "1111111", 11: "1001101", 12: "1001100", 13: "111111011",
                   14: "111111010", 16: "100111"},
    "silence": {0: "11", 1: "000", 2: "100", 3: "00101", 4: "01", 5: "0011", 6: "101101", 8: "10111",
                9: "00100", 10: "101100", 11: "1010"},
    "speech": {0: "11", 1: "1001", 2: "101", 3: "100011", 4: "00", 5: "0100", 6: "100000", 7: "0101",
               8: "0111", 9: "01101", 10: "100001", 11: "1000101", 12: "0110000", 16: "011001",
               17: "1000100", 32: "0110001"},
    "tonal": {0: "0", 1: "11110", 2: "110", 3: "1111101", 4: "101", 5: "1001011", 6: "11111101", 7: "1000",
              8: "1110", 9: "1111111", 10: "10010100", 16: "10011", 17: "1111100", 18: "11111100",
              32: "100100", 64: "10010101"},
}
ESCAPE = {"percussive": 16, "silence": 11, "speech": 7, "tonal": 7}
