"""Wire format of keys and compact ciphertexts (include/gsw_hip.h).  The reference has no
serialisation, so there is nothing of its own to pin against: the committed golden objects were
written by an independent numpy restatement of the documented layout (tests/wire_format.py,
tests/golden/make_wire_golden.py) and the library must read and reproduce them byte for byte.
"""
import os

import numpy as np
import pytest

import wire_format as wf

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P = 2147483647


def _golden(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


def test_golden_objects_parse_and_digest_matches_the_oracle(oracle):
    for name, kind, shape in (("wire_ct_fp_n2.bin", wf.CIPHERTEXT, (3, 93, 3)), ("wire_pk_fp_n2.bin", wf.PUBLIC_KEY, (1, 62, 3)),
                              ("wire_sk_fp_n2.bin", wf.SECRET_KEY, (1, 1, 2)), ("wire_ct_z32_n3.bin", wf.CIPHERTEXT, (2, 128, 4))):
        data = _golden(name)
        head, values = wf.read(data)
        assert head["kind"] == kind and values.shape == shape
        assert int.from_bytes(data[-8:], "little") == oracle.digest(values.reshape(-1))
        assert wf.write(head["kind"], head["ring_kind"], head["bits"], head["n"], head["m"], values) == data


def test_golden_files_are_what_the_script_writes():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_wire_golden", os.path.join(GOLDEN, "make_wire_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, data in mod.payloads():
        assert data == _golden(name), name


@pytest.mark.gpu
def test_library_reads_and_reproduces_the_golden_objects():
    import tfhe_amd

    gsw = tfhe_amd.GSW(tfhe_amd.Fp(), 2, 62, seed=1)
    data = _golden("wire_ct_fp_n2.bin")
    ct = tfhe_amd.Ciphertext.from_bytes(gsw.engine, data)
    assert ct.batch == 3
    assert np.array_equal(ct.to_compact(), wf.read(data)[1])
    assert ct.to_bytes() == data
    pk = gsw.pk_from_bytes(_golden("wire_pk_fp_n2.bin"))
    assert np.array_equal(pk.pk_matrix, wf.read(_golden("wire_pk_fp_n2.bin"))[1][0])
    assert gsw.pk_to_bytes(pk) == _golden("wire_pk_fp_n2.bin")
    sk = gsw.sk_from_bytes(_golden("wire_sk_fp_n2.bin"))
    assert np.array_equal(sk.t, wf.read(_golden("wire_sk_fp_n2.bin"))[1][0, 0])
    assert sk.s[0] == 1 and np.array_equal((sk.s[1:].astype(np.uint64) + sk.t) % P, np.zeros(2))
    assert gsw.sk_to_bytes(sk) == _golden("wire_sk_fp_n2.bin")
    wide = tfhe_amd.GSW(tfhe_amd.Zpow2(32), 3, 96, seed=1, check_max=False)
    data = _golden("wire_ct_z32_n3.bin")
    assert tfhe_amd.Ciphertext.from_bytes(wide.engine, data).to_bytes() == data


@pytest.mark.gpu
def test_keys_and_ciphertexts_survive_the_wire():
    """keygen on one context, everything shipped as bytes to a second one: encrypt there, multiply,
    ship the product back, decrypt with the shipped secret key."""
    import tfhe_amd

    a = tfhe_amd.GSW(tfhe_amd.Fp(), 10, 310, seed=21)
    b = tfhe_amd.GSW(tfhe_amd.Fp(), 10, 310, seed=22)
    sk, pk = a.keygen()
    pk_b = b.pk_from_bytes(a.pk_to_bytes(pk))
    c1, c2 = b.encrypt_batch(pk_b, [0, 1, 1]), b.encrypt_batch(pk_b, [1, 1, 0])
    prod_bytes = b.nand(c1, c2).to_bytes()
    back = tfhe_amd.Ciphertext.from_bytes(a.engine, prod_bytes)
    assert a.decrypt_batch(a.sk_from_bytes(a.sk_to_bytes(sk)), back).tolist() == [1, 0, 1]


@pytest.mark.gpu
def test_reader_rejects_damaged_or_foreign_objects():
    import tfhe_amd

    gsw = tfhe_amd.GSW(tfhe_amd.Fp(), 2, 62, seed=1)
    good = bytearray(_golden("wire_ct_fp_n2.bin"))

    def rejects(data, needle):
        with pytest.raises(tfhe_amd.GswError, match=needle):
            tfhe_amd.Ciphertext.from_bytes(gsw.engine, bytes(data))

    rejects(b"GSW0" + good[4:], "magic")
    rejects(good[:-9] + good[-8:], "bytes")
    flipped = bytearray(good)
    flipped[40] ^= 1
    rejects(flipped, "digest")
    big = bytearray(good)
    big[32:36] = (P).to_bytes(4, "little")  # p itself is not canonical
    rejects(big, "canonical")
    rejects(_golden("wire_pk_fp_n2.bin"), "kind")
    rejects(_golden("wire_ct_z32_n3.bin"), "ring")
    other = tfhe_amd.GSW(tfhe_amd.Fp(), 3, 93, seed=1)
    with pytest.raises(tfhe_amd.GswError, match="n, m"):
        tfhe_amd.Ciphertext.from_bytes(other.engine, bytes(good))
    with pytest.raises(tfhe_amd.GswError, match="kind"):
        gsw.pk_from_bytes(_golden("wire_sk_fp_n2.bin"))
    empty = tfhe_amd.Ciphertext(gsw.engine, 1)
    empty.batch = 0
    assert wf.read(empty.to_bytes())[1].shape == (0, 93, 3)
