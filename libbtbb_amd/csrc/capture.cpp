// capture.cpp -- BR/EDR capture files downstream of the path: classic pcap with
// LINKTYPE_BLUETOOTH_BREDR_BB (255) and pcapng with the same link type.
//
// Replaces, for BR/EDR only, lib/src/pcap.c:48-228 (btbb_pcap_*), lib/src/pcapng.c:35-322 (the
// generic section / interface / packet block writer) and lib/src/pcapng-bt.c:32-343
// (btbb_pcapng_*).  Pure host file I/O over fields the GPU decode left in the packet object; the
// on-disk layouts are pcap-common.h:62-96 and pcapng.h:29-147, pcapng-bt.h:28-73.
//
// Own design, same bytes: the reference maps the two header blocks of a pcapng file with mmap and
// edits them in place; here they are kept as memory images and written back with pwrite after
// every change, so a reader of the file sees the same thing at the same moments.
//
// Bytes the reference leaves undefined are written as zero here:
//  * the 0..3 padding bytes between the captured data of an enhanced packet block and its options
//    word (stack garbage of pcapng_bredr_packet, pcapng-bt.c:197-226);
//  * the 4 bytes that pcapng_append_interface_option copies beyond the BD_ADDR / clock option
//    structs, whose option_length counts the option header a second time (pcapng-bt.c:274-277,
//    314-317; pcapng.c:272-274).
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/stat.h>
#include <vector>
#include "packet_obj.h"
#include "../../include/btbb.h"

namespace {

constexpr uint32_t DLT_BREDR_BB = 255;          // pcap-common.h:76-78
constexpr uint32_t MAX_PAYLOAD = 400;           // pcap-common.h:75
constexpr uint16_t F_DEWHITENED = 0x0001, F_SIGPOWER = 0x0002, F_NOISEPOWER = 0x0004, F_REFLAP = 0x0010,
		   F_PAYLOAD = 0x0020, F_REFUAP = 0x0080;        // pcap-common.h:59-73

void put16(std::vector<uint8_t> &b, size_t at, uint16_t v) { b[at] = (uint8_t)v; b[at + 1] = (uint8_t)(v >> 8); }
void put32(std::vector<uint8_t> &b, size_t at, uint32_t v) { put16(b, at, (uint16_t)v); put16(b, at + 2, (uint16_t)(v >> 16)); }
void put64(std::vector<uint8_t> &b, size_t at, uint64_t v) { put32(b, at, (uint32_t)v); put32(b, at + 4, (uint32_t)(v >> 32)); }

// the 22-byte LINKTYPE_BLUETOOTH_BREDR_BB pseudo header + payload (pcap-common.h:80-96), little endian
std::vector<uint8_t> bredr_record(const btbb_packet *pkt, int8_t sigdbm, int8_t noisedbm, uint32_t reflap, uint8_t refuap)
{
	uint16_t flags = F_DEWHITENED | F_SIGPOWER | (noisedbm < sigdbm ? F_NOISEPOWER : 0) |
			 (reflap != LAP_ANY ? F_REFLAP : 0) | (refuap != UAP_ANY ? F_REFUAP : 0);
	const int len = btbb_packet_get_payload_length(pkt);
	std::vector<char> bytes((size_t)(len > 0 ? len : 1));
	btbb_get_payload_packed(pkt, bytes.data());
	const uint32_t caplen = len < 0 ? 0 : ((uint32_t)len < MAX_PAYLOAD ? (uint32_t)len : MAX_PAYLOAD);
	if (caplen)
		flags |= F_PAYLOAD;
	std::vector<uint8_t> r(22 + caplen, 0);
	r[0] = btbb_packet_get_channel(pkt);
	r[1] = (uint8_t)sigdbm;
	r[2] = (uint8_t)noisedbm;
	r[3] = btbb_packet_get_ac_errors(pkt);
	r[4] = (uint8_t)((btbb_packet_get_transport(pkt) << 4) | btbb_packet_get_modulation(pkt));
	r[5] = 0;                                   // corrected header bits: TODO in the reference too
	put16(r, 6, 0);                             // corrected payload bits: likewise
	put32(r, 8, btbb_packet_get_lap(pkt));
	put32(r, 12, (reflap & 0xffffff) | ((uint32_t)refuap << 24));
	put32(r, 16, btbb_packet_get_header_packed(pkt));
	put16(r, 20, flags);
	memcpy(r.data() + 22, bytes.data(), caplen);
	return r;
}

bool write_all(int fd, const void *p, size_t n)
{
	const char *c = (const char *)p;
	while (n) {
		ssize_t k = write(fd, c, n);
		if (k < 0) {
			if (errno == EINTR)
				continue;
			return false;
		}
		c += k;
		n -= (size_t)k;
	}
	return true;
}

bool pwrite_all(int fd, const void *p, size_t n, off_t at)
{
	const char *c = (const char *)p;
	while (n) {
		ssize_t k = pwrite(fd, c, n, at);
		if (k < 0) {
			if (errno == EINTR)
				continue;
			return false;
		}
		c += k;
		n -= (size_t)k;
		at += k;
	}
	return true;
}

} // namespace

// ---- classic pcap ---------------------------------------------------------------------------
struct btbb_pcap_handle {
	FILE *file;
};

// result codes: pcap.c:32-37 and pcapng.h:163-172, returned negated
enum { CAP_OK = 0, CAP_INVALID_HANDLE = 1, CAP_FILE_NOT_ALLOWED = 2, NG_NO_MEMORY = 5, NG_WRITE_ERROR = 6 };
enum { PCAP_NO_MEMORY = 3 };

// ---- pcapng -----------------------------------------------------------------------------------
struct btbb_pcapng_handle {
	int fd;
	std::vector<uint8_t> section;        // section header block, page padded
	size_t next_section_option;
	std::vector<uint8_t> interface;      // interface description block, page padded
	size_t next_interface_option;
};

namespace {

// append one option (code, length, value) + zero padding to a multiple of 4
void push_option(std::vector<uint8_t> &b, uint16_t code, const void *value, uint16_t length)
{
	const size_t at = b.size();
	b.resize(at + 4 + 4 * (((size_t)length + 3) / 4), 0);
	put16(b, at, code);
	put16(b, at + 2, length);
	memcpy(b.data() + at + 4, value, length);
}

// pad a block image to whole pages with room for `space` more option bytes (pcapng.c:103-114),
// then close it with the padding option 0xffff, the end-of-options word and the trailing length
// (pcapng.c:186-206)
void finish_block(std::vector<uint8_t> &b, size_t next_option, size_t space, size_t page)
{
	const size_t size = page * ((b.size() + 4 + space + page - 1) / page);
	b.resize(size, 0);
	put16(b, next_option, 0xffff);
	put16(b, next_option + 2, (uint16_t)(size - next_option - 12));
	put32(b, 4, (uint32_t)size);
	put32(b, size - 4, (uint32_t)size);
}

// pcapng_append_interface_option, pcapng.c:262-289: `copy` bytes go in, of which the first
// `defined` come from the caller's struct
int add_interface_option(btbb_pcapng_handle *h, const std::vector<uint8_t> &opt, size_t copy)
{
	if (!h || h->fd == -1)
		return CAP_INVALID_HANDLE;
	if (h->interface.empty() || !h->next_interface_option ||
	    h->next_interface_option + 4 * ((copy + 3) / 4) + 12 > h->interface.size())
		return NG_NO_MEMORY;
	const size_t at = h->next_interface_option;
	memset(h->interface.data() + at, 0, copy);
	memcpy(h->interface.data() + at, opt.data(), opt.size() < copy ? opt.size() : copy);
	h->next_interface_option += 4 * ((copy + 3) / 4);
	const size_t pad = h->next_interface_option;
	put16(h->interface, pad, 0xffff);
	put16(h->interface, pad + 2, (uint16_t)(h->interface.size() - pad - 12));
	if (!pwrite_all(h->fd, h->interface.data() + at, pad + 4 - at, (off_t)(h->section.size() + at)))
		return NG_WRITE_ERROR;
	return CAP_OK;
}

} // namespace

extern "C" {

int btbb_pcap_close(btbb_pcap_handle *h)
{
	if (h && h->file)
		fclose(h->file);
	if (h) {
		free(h);
		return 0;
	}
	return -CAP_INVALID_HANDLE;
}

/* pcap.c:48-101: nanosecond-resolution pcap (magic 0xa1b23c4d), version 2.4, snaplen 400 */
int btbb_pcap_create_file(const char *filename, btbb_pcap_handle **ph)
{
	btbb_pcap_handle *h = (btbb_pcap_handle *)calloc(1, sizeof(*h));
	if (!h)
		return -PCAP_NO_MEMORY;
	h->file = fopen(filename, "w");
	if (!h->file) {
		perror("PCAP error:");
		btbb_pcap_close(h);
		return -CAP_FILE_NOT_ALLOWED;
	}
	std::vector<uint8_t> hdr(24, 0);
	put32(hdr, 0, 0xa1b23c4d);
	put16(hdr, 4, 2);
	put16(hdr, 6, 4);
	put32(hdr, 16, MAX_PAYLOAD);
	put32(hdr, 20, DLT_BREDR_BB);
	fwrite(hdr.data(), hdr.size(), 1, h->file);
	*ph = h;
	return 0;
}

/* pcap.c:168-210 */
int btbb_pcap_append_packet(btbb_pcap_handle *h, const uint64_t ns, const int8_t sigdbm, const int8_t noisedbm,
			    const uint32_t reflap, const uint8_t refuap, const btbb_packet *pkt)
{
	if (!h || !h->file)
		return -CAP_INVALID_HANDLE;
	const std::vector<uint8_t> rec = bredr_record(pkt, sigdbm, noisedbm, reflap, refuap);
	std::vector<uint8_t> hdr(16, 0);
	put32(hdr, 0, (uint32_t)(ns / 1000000000ull));
	put32(hdr, 4, (uint32_t)(ns % 1000000000ull));
	put32(hdr, 8, (uint32_t)rec.size());
	put32(hdr, 12, (uint32_t)rec.size());
	fwrite(hdr.data(), hdr.size(), 1, h->file);
	fwrite(rec.data(), rec.size(), 1, h->file);
	fflush(h->file);
	return 0;
}

/* pcapng-bt.c:335-343: closes and, like the reference, always reports -PCAPNG_INVALID_HANDLE */
int btbb_pcapng_close(btbb_pcapng_handle *h)
{
	if (h) {
		if (h->fd != -1)
			close(h->fd);
		delete h;
	}
	return -CAP_INVALID_HANDLE;
}

/* pcapng-bt.c:96-157 on top of pcapng.c:35-217: section header with the user-application option
 * "libbtbb", one interface (link type 255, snaplen 400, optional if_description, if_tsresol 9),
 * each block padded to whole pages so that options can be added later */
int btbb_pcapng_create_file(const char *filename, const char *interface_desc, btbb_pcapng_handle **ph)
{
	const size_t page = (size_t)sysconf(_SC_PAGESIZE);
	const int fd = open(filename, O_RDWR | O_CREAT | O_EXCL, S_IRUSR | S_IWUSR | S_IRGRP | S_IWGRP);
	if (fd == -1) {
		// pcapng.c:57-72 maps errno to FILE_EXISTS / TOO_MANY_FILES_OPEN / NO_MEMORY /
		// FILE_NOT_ALLOWED, but :100-102 then overwrites every one of them with
		// FILE_WRITE_ERROR because nothing has been written yet; callers see -6
		return -NG_WRITE_ERROR;
	}
	btbb_pcapng_handle *h = new btbb_pcapng_handle();
	h->fd = fd;

	// section header block: type, length, byte-order magic, version 1.0, section length
	h->section.assign(24, 0);
	put32(h->section, 0, 0x0a0d0d0a);
	put32(h->section, 8, 0x1a2b3c4d);
	put16(h->section, 12, 1);
	push_option(h->section, 4 /* shb_userappl */, "libbtbb", 7);
	h->next_section_option = h->section.size();
	finish_block(h->section, h->next_section_option, page, page);

	// interface description block
	h->interface.assign(16, 0);
	put32(h->interface, 0, 1);
	put16(h->interface, 8, (uint16_t)DLT_BREDR_BB);
	put32(h->interface, 12, MAX_PAYLOAD);
	if (interface_desc) {
		char desc[256];
		strncpy(desc, interface_desc, 256);
		desc[255] = '\0';
		// a zero-length description ends the reference's option walk: nothing is written
		if (desc[0])
			push_option(h->interface, 3 /* if_description */, desc, (uint16_t)strlen(desc));
	}
	h->next_interface_option = h->interface.size();
	finish_block(h->interface, h->next_interface_option, page, page);
	put64(h->section, 16, (uint64_t)h->interface.size());     // section length so far

	if (!write_all(fd, h->section.data(), h->section.size()) ||
	    !write_all(fd, h->interface.data(), h->interface.size())) {
		btbb_pcapng_close(h);
		return -NG_WRITE_ERROR;
	}
	// nanosecond timestamps (check_and_fix_tsresol, pcapng-bt.c:56-91)
	const uint8_t resol = 9;
	std::vector<uint8_t> opt(5, 0);
	put16(opt, 0, 9 /* if_tsresol */);
	put16(opt, 2, 1);
	opt[4] = resol;
	const int rc = add_interface_option(h, opt, 5);
	if (rc) {
		btbb_pcapng_close(h);
		return -rc;
	}
	*ph = h;
	return 0;
}

/* pcapng-bt.c:166-257 + pcapng.c:291-309: one enhanced packet block, then the section length in
 * the header grows by the block size */
int btbb_pcapng_append_packet(btbb_pcapng_handle *h, const uint64_t ns, const int8_t sigdbm, const int8_t noisedbm,
			      const uint32_t reflap, const uint8_t refuap, const btbb_packet *pkt)
{
	if (!h || h->fd == -1)
		return -CAP_INVALID_HANDLE;
	const std::vector<uint8_t> rec = bredr_record(pkt, sigdbm, noisedbm, reflap, refuap);
	const uint32_t block_length = 4 * ((36 + (uint32_t)rec.size() + 3) / 4);
	std::vector<uint8_t> blk(block_length, 0);
	put32(blk, 0, 6);                               // enhanced packet block
	put32(blk, 4, block_length);
	put32(blk, 8, 0);                               // interface 0
	put32(blk, 12, (uint32_t)(ns >> 32));
	put32(blk, 16, (uint32_t)ns);
	put32(blk, 20, (uint32_t)rec.size());
	put32(blk, 24, (uint32_t)rec.size());
	memcpy(blk.data() + 28, rec.data(), rec.size());
	put32(blk, block_length - 4, block_length);     // the word before it stays 0: no options
	if (!write_all(h->fd, blk.data(), blk.size()))
		return 0;                               // the reference drops the write error (pcapng.c:299-301)
	uint64_t section_length = 0;
	for (int i = 7; i >= 0; i--)
		section_length = (section_length << 8) | h->section[16 + i];
	put64(h->section, 16, section_length + block_length);
	pwrite_all(h->fd, h->section.data() + 16, 8, 16);
	return 0;
}

/* pcapng-bt.c:259-292: interface option 0xd340 {bd_addr[6], uap_mask, nap_valid} */
int btbb_pcapng_record_bdaddr(btbb_pcapng_handle *h, const uint64_t bdaddr, const uint8_t uapmask, const uint8_t napvalid)
{
	std::vector<uint8_t> opt(12, 0);
	put16(opt, 0, 0xd340);
	put16(opt, 2, 12);                              // the reference counts the option header too
	for (int i = 0; i < 6; i++)
		opt[4 + i] = (uint8_t)(bdaddr >> (8 * i));
	opt[10] = uapmask;
	opt[11] = napvalid;
	return -add_interface_option(h, opt, 4 + 12);
}

/* pcapng-bt.c:294-333: interface option 0xd341 {ts, lap_uap, clk, clk_mask} */
int btbb_pcapng_record_btclock(btbb_pcapng_handle *h, const uint64_t bdaddr, const uint64_t ns, const uint32_t clk,
			       const uint32_t clkmask)
{
	std::vector<uint8_t> opt(24, 0);
	put16(opt, 0, 0xd341);
	put16(opt, 2, 24);
	put64(opt, 4, ns);
	put32(opt, 12, (uint32_t)(bdaddr & 0xffffffff));
	put32(opt, 16, clk);
	put32(opt, 20, clkmask);
	return -add_interface_option(h, opt, 4 + 24);
}

} // extern "C"
