"""dcase2019_task4_amd - MI355X-native mean-teacher CRNN training path for DCASE 2019 task 4.

Host-side mirrors of the reference's interfaces for ONE hot path:
  crnn.CRNN                      <- baseline/models/CRNN.py
  train.MeanTeacherStep / train  <- baseline/main.py train(), update_ema_variables()
  features.*                     <- DatasetDcase2019Task4.calculate_mel_spec, DataLoad transforms
  dist.*                         <- data-parallel sharding of the step (new capability)
All arithmetic runs in hand-written gfx950 HIP kernels behind the C-ABI of include/dcase_sed.h.
"""
from ._lib import SedError, lib, LIB_PATH  # noqa: F401
