"""placeholder: the reference imports the name at module level; nothing on the message-passing path calls into it"""
