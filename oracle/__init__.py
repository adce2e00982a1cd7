"""CPU oracle for the SatFlow spatiotemporal hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``satflow_amd/`` imports it, and the
product path raises when the HIP library is missing instead of falling back to
anything here.

What it is: a plain-PyTorch (CPU, fp32) functional restatement of the
reference algorithm for the path named by ``BASELINE.json: north_star``.

* ``oracle.convlstm``  -- ConvLSTMCell / ConvLSTM encoder-decoder.  Parity is
  PINNED: checked bit-for-bit-close against the reference itself imported in
  the build container (``tests/golden/make_golden.py``), the resulting tensors
  are committed under ``tests/golden/``.
* ``oracle.metnet``    -- MetNet stack (preprocessor, ConditionTime,
  TimeDistributed(DownSampler), ConvGRU, axial attention, 1x1 head).
  ConditionTime / TimeDistributed / space_to_depth are pinned against the
  in-tree reference copies.  The rest lives in the un-vendored third-party
  packages ``metnet>=0.0.3`` and ``axial_attention`` (reference
  ``requirements.txt:18``), neither importable here: for those blocks the
  oracle follows the published upstream algorithm and PARITY IS UNPINNED
  except for the reference's own shape/NaN test (``tests/test_models.py:42-61``).
"""
