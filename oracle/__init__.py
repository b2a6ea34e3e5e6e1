"""TEST INFRASTRUCTURE ONLY.

CPU restatement (torch-CPU / numpy) of the DCASE2019 task-4 baseline hot path, used as the
parity checker for the HIP implementation in ``dcase2019_task4_amd``.  Nothing in the product
package may import from here: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do.

Pinning status
--------------
* model / losses / optimiser / EMA (``ref_cpu.py``): PINNED against golden vectors captured from
  the imported reference (``gen_golden.py`` -> ``tests/golden/*.npz``).
* ``Scaler`` statistics / ``sigmoid_rampup`` / pad-trunc / normalise (``features_np.py``): PINNED
  the same way (those reference files import here).
* librosa-dependent arithmetic (``stft``, ``filters.mel``, ``amplitude_to_db``): **parity
  unpinned** - librosa is an unpinned third-party dependency of the reference
  (environment.yml:17 bare ``librosa``), absent from /root/reference and from this image; the
  restatement follows librosa's published algorithm and is cross-checked against ``torch.stft``
  and ``numpy.fft`` only.
"""
