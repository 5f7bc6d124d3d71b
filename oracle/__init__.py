"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's audio-tagging inference path
(waveform -> STFT/log-mel -> bn0 -> ConvNeXt-Tiny -> logits/probs/embeddings).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from here, and only as the *checker* (or as the
timed CPU baseline) -- never as the product path.  The product
(``audioset-convnext-inf_amd``) must not import this package and fails loudly when
its HIP library is missing.

Parity pin status
-----------------
* Backbone (bn0 onward; reference ``convnext.py:269-331``): PINNED.  ``ref_cpu.py``
  is checked against the reference's own ``ConvNeXt`` class imported in the build
  container (``tests/golden/make_goldens.py``) and against the committed golden
  vectors generated from it (``tests/test_oracle.py``).
* Frontend (torchlibrosa 0.0.9 ``Spectrogram`` / ``LogmelFilterBank`` on top of
  librosa 0.8.1; ``environment.yml:48,71``): the library is a third-party dependency
  that is absent from ``/root/reference`` and from this image, and the reference holds
  no tests or vectors at that boundary.  ``torchlibrosa_spec.py`` restates the
  published algorithm; the reference's *call sites* (``convnext.py:179-200,298-299``)
  are exercised through it.  => frontend constants are "parity unpinned" until a
  real checkpoint (which stores the library's tables as buffers) is supplied; see
  DESIGN.md "Oracle".
"""
