"""Test-side access to the checker (oracle/): the CPU restatement and, where present, the compiled
reference (oracle/_ref).  Imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by longqc_amd/."""
import os
import subprocess
from typing import List, Optional

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_CLI = os.path.join(ORACLE_DIR, "lqcov_oracle")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "minimap2-coverage")
REF_HARNESS = os.path.join(ORACLE_DIR, "_ref", "ref_harness")


def ensure_oracle():
    if not os.path.exists(ORACLE_CLI):
        subprocess.run(["make", "-C", ORACLE_DIR, "oracle"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return ORACLE_CLI


def have_ref() -> bool:
    return os.path.exists(REF_BIN) and os.path.exists(REF_HARNESS)


def table(argv: List[str], extra: Optional[List[str]] = None) -> str:
    """9-column table of the CPU restatement for a minimap2-coverage argv."""
    ensure_oracle()
    r = subprocess.run([ORACLE_CLI, "table"] + (extra or []) + [str(a) for a in argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle failed: " + r.stderr[-2000:])
    return r.stdout


def dump(what: str, opts: List[str], files: List[str]) -> str:
    ensure_oracle()
    r = subprocess.run([ORACLE_CLI, what] + opts + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle failed: " + r.stderr[-2000:])
    return r.stdout


def ref_table(argv: List[str]) -> str:
    r = subprocess.run([REF_BIN] + [str(a) for a in argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("reference failed (%d): %s" % (r.returncode, r.stderr[-2000:]))
    return r.stdout


def ref_dump(args: List[str]) -> str:
    r = subprocess.run([REF_HARNESS] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("ref_harness failed: " + r.stderr[-2000:])
    return r.stdout
