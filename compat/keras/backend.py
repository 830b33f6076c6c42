"""K.set_image_dim_ordering('tf') (train_2ddense.py:18) -- the MI355X path is channels-last at the API by construction"""


def set_image_dim_ordering(order):
    if order != "tf":
        raise ValueError("only channels-last ('tf') ordering is supported, as in the reference scripts")


def image_dim_ordering():
    return "tf"


def clear_session():
    pass
