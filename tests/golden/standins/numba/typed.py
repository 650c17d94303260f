class List(list):
    @classmethod
    def empty_list(cls, _type=None):
        return cls()


class Dict(dict):
    @classmethod
    def empty(cls, key_type=None, value_type=None):
        return cls()
