"""Generate tests/golden/text_golden.json from the REFERENCE's own text-target code (build container only):
processing/target_normalizers.py::aurora4_normalizer and processing/target_coder.py::TextCoder, converted with
lib2to3 into a scratch directory under /tmp exactly as oracle/make_golden_io.py does.  The fixture holds inputs and
expected outputs only (transcriptions, their normalised form, their label vectors, the alphabet).

    python oracle/make_golden_text.py        # needs /root/reference
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import warnings

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")

TRANSCRIPTIONS = [
    "THE QUICK BROWN FOX",
    "HELLO ,COMMA WORLD .PERIOD",
    "IT'S A \"DOUBLE-QUOTE TEST ?QUESTION-MARK",
    "<NOISE> NOISY START AND END <NOISE>",
    "NUMBERS 123 AND SYMBOLS % $ ARE UNKNOWN",
    "MR. O'NEIL -HYPHEN SMITH -DASH WENT (LEFT-PAREN HOME )RIGHT-PAREN",
    "ELLIPSIS ...ELLIPSIS AND ;SEMI-COLON :COLON /SLASH",
    "A",
    "",
    "MIXED Case stays lower",
]


def main():
    scratch = tempfile.mkdtemp(prefix="tfkaldi_ref_txt_", dir="/tmp")
    dst = os.path.join(scratch, "processing")
    shutil.copytree(os.path.join(REF, "processing"), dst, ignore=shutil.ignore_patterns("__pycache__"))
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", os.path.join(dst, "target_coder.py"),
                           os.path.join(dst, "target_normalizers.py")], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    sys.path.insert(0, dst)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # `is not ' '` in the reference raises a SyntaxWarning on Python 3
        import target_coder as tc
        import target_normalizers as tn
    coder = tc.TextCoder(tn.aurora4_normalizer)
    cases = []
    for text in TRANSCRIPTIONS:
        cases.append({"text": text, "normalized": tn.aurora4_normalizer(text, list(coder.lookup.keys())),
                      "encoded": [int(x) for x in coder.encode(text)]})
    alphabet = [sym for sym, _ in sorted(coder.lookup.items(), key=lambda kv: kv[1])]
    # (decode is left out: the reference indexes `lookup.keys()`, whose order is arbitrary under Python 2)
    out = {"alphabet": alphabet, "num_labels": int(coder.num_labels), "cases": cases}
    with open(os.path.join(GOLD, "text_golden.json"), "w") as fid:
        json.dump(out, fid, indent=1)
    shutil.rmtree(scratch)
    print("wrote %d cases" % len(cases))


if __name__ == "__main__":
    main()
