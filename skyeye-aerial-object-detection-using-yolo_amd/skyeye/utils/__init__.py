"""skyeye.utils -- only the pieces on the inference hot path (the reference's package cannot be imported at all:
utils/__init__.py:10,22-25 pull in cv2 and an undefined ``fuse_conv_and_bn``)."""
